"""Two ranks on the one GPU of the test box (gloo between them, MG_SHARE_GPU rehearsal of the
multi-GPU path): time-sharded single-assay mode (flat-field max all-reduce + bead-table broadcast) and
the weak-scaling marker-table gather against the single-process results."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, C, H, W = 4, 2, 384, 512


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stack():
    from magnify_amd.stack import synthetic_stack

    return synthetic_stack(T, C, H, W, seed=321, beads_per_mpx=300.0)[0]


def _worker(rank, world, port, ret, save_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MG_SHARE_GPU="1", MG_DIST_BACKEND="gloo")
    import torch

    from magnify_amd import distributed as mgd
    from magnify_amd.stack import StackProcessor

    r, w, _ = mgd.init_from_env()
    stack = _stack()  # same seed on every rank: the full stack, of which a rank uses its shard
    lo, hi = mgd.shard_range(T, r, w)
    proc = StackProcessor(hi - lo, C, H, W, num_iter=60000, search_channels=(0,), mode="R")
    out = mgd.run_mode_r(proc, stack[lo:hi].contiguous(), 0.9, 100.0, seed=4)
    ret[rank] = (out["beads"][0].copy(), out["roi"].cpu().numpy().copy(), out["sums"].cpu().numpy().copy(),
                 proc.image.cpu().numpy().copy(), (lo, hi))
    # every rank saves ITS shard of the result from device memory (ROI pixels stay sharded by GPU)
    import magnify_amd as mg

    ds = mg.Dataset(attrs={"name": "mode R", "first_timepoint": lo})
    ds["roi"] = mg.DataArray(out["roi"], ("mark", "channel", "time", "roi_y", "roi_x"))
    ds = ds.assign_coords(fg=(("mark", "roi_y", "roi_x"), out["fg"].bool()), time=(("time",), np.arange(lo, hi)))
    mg.save(os.path.join(save_dir, f"shard_rank{rank}.nc"), ds)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_mode_r_time_sharded_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    from magnify_amd import hotpath
    from magnify_amd.stack import StackProcessor

    hotpath.require_gpu()
    stack = _stack()
    proc = StackProcessor(T, C, H, W, num_iter=60000, search_channels=(0,), mode="R")
    want = proc(stack, 0.9, 100.0, seed=4)
    assert len(want["beads"][0]) >= 20
    world, port = 2, _free_port()
    manager = mp.get_context("spawn").Manager()
    ret = manager.dict()
    mp.spawn(_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    import magnify_amd as mg

    shards = [mg.load(tmp_path / f"shard_rank{r}.nc") for r in range(world)]
    np.testing.assert_array_equal(np.concatenate([s["roi"].values for s in shards], axis=2), want["roi"].cpu().numpy())
    np.testing.assert_array_equal(shards[1].coords["fg"].values, want["fg"].cpu().numpy().astype(bool))
    assert [int(s.attrs["first_timepoint"]) for s in shards] == [0, 2] and list(shards[1].coords["time"].values) == [2, 3]
    for rank in range(world):
        beads, roi, sums, image, (lo, hi) = ret[rank]
        np.testing.assert_array_equal(beads, want["beads"][0])  # broadcast from the owner of time 0
        np.testing.assert_array_equal(image, proc.image[lo:hi].cpu().numpy())  # global maxima on every shard
        np.testing.assert_array_equal(roi, want["roi"][:, :, lo:hi].cpu().numpy())
        np.testing.assert_array_equal(sums, want["sums"][:, :, lo:hi].cpu().numpy())


def test_rccl_collectives_on_one_rank():
    """Backend "nccl" IS RCCL on ROCm.  The one-GPU box cannot hold two RCCL ranks, but a process group of ONE rank is
    legal: the collectives of the multi-GPU path (marker-table all-gather, flat-field max all-reduce, bead broadcast,
    and mode R on top of them) run through RCCL on device tensors here, in a child process that initialises the
    group before anything else touches the GPU."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MG_SHARE_GPU", "MG_DIST_BACKEND")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py")], env=env, capture_output=True, text=True,
                          timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    rec = json.loads([ln for ln in done.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["table_on_device"] and rec["table_equal"]
    assert rec["max"] == [3.5, -1.0] and rec["shared"] == [[1, 2, 3], [4, 5, 6]] and rec["empty_shape"] == [0, 3]
    assert rec["mode_r_beads"] >= 5 and rec["mode_r_equal"]


def _bench(*flags):
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--size", "512", "--num-iter", "40000",
           "--no-cpu", *flags]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no torchrun): the parent spawns the ranks; on this one-GPU box
    they share cuda:0 over gloo.  Weak scaling carries the strong-scaling object; `--scaling strong`
    processes ONE stack, and its shards find exactly the markers of the single-rank run (an assay's
    RNG stream and synthetic content depend on its global index only for shard starts that coincide)."""
    weak = _bench("--gpus", "2", "--timepoints", "4")
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["ranks"]["world_size"] == 2
    assert weak["ranks"]["launcher"] == "self" and weak["ranks"]["shared_gpu"] is True
    assert weak["config"]["timepoints_per_gpu"] == 4 and weak["config"]["timepoints_total"] == 8
    assert weak["strong"]["timepoints_per_gpu"] == 2 and weak["strong"]["value"] > 0
    assert abs(weak["value"] - 8 * 4 * 512 * 512 / 1e6 / (weak["ms_per_step"] / 1e3)) < 1e-6 * weak["value"]
    strong = _bench("--gpus", "2", "--timepoints", "4", "--scaling", "strong")
    assert strong["scaling"] == "strong" and strong["config"]["timepoints_per_gpu"] == 2
    assert strong["config"]["timepoints_total"] == 4
    assert abs(strong["value"] - 4 * 4 * 512 * 512 / 1e6 / (strong["ms_per_step"] / 1e3)) < 1e-6 * strong["value"]
    one = _bench("--gpus", "1", "--timepoints", "2")
    assert one["n_gpus"] == 1 and one["strong"] is None and one["ranks"]["world_size"] == 1
    assert one["roofline"]["frac"] is not None and "cpu_baseline" in one


@pytest.mark.gpu
def test_marker_table_kernel_equals_tensor_packing():
    """mg_marker_table (one kernel over the device-resident bead tables, counts and sums) against the plain tensor
    packing of the same result: a StackProcessor result (padded bead rows + device offsets, an assay without beads in
    the middle), a result with host bead lists only, and an empty one."""
    import numpy as np
    import torch

    from magnify_amd import distributed as mgd
    from magnify_amd import hotpath as hp
    from magnify_amd.stack import StackProcessor, synthetic_stack

    hp.require_gpu()
    stack = synthetic_stack(3, 2, 256, 320, seed=11, beads_per_mpx=400.0)[0]
    stack[1] = 100  # a timepoint without beads
    proc = StackProcessor(3, 2, 256, 320, num_iter=40000, search_channels=(0,), mode="P")
    for call in range(2):  # the checked call and an optimistic one (ROI pass queued before the counts are known)
        out = proc(stack, 0.9, 100.0, seed=4)
        assert len(out["beads"][1]) == 0 and len(out["beads"][0]) > 0
        got = mgd.marker_table(out, 7, 2, torch.device("cuda"))
        host = {"beads": out["beads"], "counts": out["counts"].cpu(), "sums": out["sums"].cpu()}
        want = mgd.marker_table(host, 7, 2, torch.device("cpu"))
        assert got.is_cuda and torch.equal(got.cpu(), want)
        only_lists = {"beads": out["beads"], "counts": out["counts"], "sums": out["sums"]}
        assert torch.equal(mgd.marker_table(only_lists, 7, 2, torch.device("cuda")).cpu(), want)
    empty = {"beads": [np.empty((0, 3), np.int32)], "counts": torch.empty((0, 2), dtype=torch.int32, device="cuda"),
             "sums": torch.empty((0, 2, 1, 2), dtype=torch.float64, device="cuda")}
    assert mgd.marker_table(empty, 0, 2, torch.device("cuda")).shape == (0, 10)


# ---- config C5 across ranks: one streamed series, its time axis split over two ranks -----------------------------
C5 = dict(n_t=6, n_c=2, grid=2, tile=160, overlap=16, chunk=2, num_iter=40000)


def _c5_series(root):
    """One OME-BigTIFF per tile position with (time, channel) pages, cut from a synthetic canvas."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tiffwrite import ome_xml, write_tiff

    from magnify_amd.stack import synthetic_stack

    n_t, n_c, g, ty, ov = C5["n_t"], C5["n_c"], C5["grid"], C5["tile"], C5["overlap"]
    step = ty - 2 * (ov // 2) - ov % 2
    side = (g - 1) * step + ty
    canvas = synthetic_stack(n_t, n_c, side, side, seed=77, beads_per_mpx=400.0)[0].cpu().numpy()
    for r in range(g):
        for c in range(g):
            pages = [canvas[t, ch, r * step: r * step + ty, c * step: c * step + ty] for t in range(n_t) for ch in range(n_c)]
            write_tiff(os.path.join(root, f"acq_r{r}_c{c}.ome.tif"), pages, bigtiff=True,
                       description=ome_xml(size_c=n_c, size_t=n_t, size_y=ty, size_x=ty, channel_names=["a", "b"]))
    return os.path.join(root, "acq_r(row)_c(col).ome.tif")


def _c5_flat():
    ty = C5["tile"]
    yy, xx = np.mgrid[0:ty, 0:ty]
    return (1 - 0.15 * (((yy - (ty - 1) / 2) / (ty / 2)) ** 2 + ((xx - (ty - 1) / 2) / (ty / 2)) ** 2)).astype(np.float32)


def _c5_worker(rank, world, port, pattern, save_dir, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MG_SHARE_GPU="1", MG_DIST_BACKEND="gloo")
    import torch

    import magnify_amd as mg
    from magnify_amd import distributed as mgd

    r, w, _ = mgd.init_from_env()
    sink = mg.SaveSink(os.path.join(save_dir, "t{index:03d}.nc"))  # ONE pattern for all ranks: {index} is global
    firsts = []
    table, (lo, hi) = mgd.stream_series(pattern, C5["chunk"], _c5_flat(), 90.0, seed=9, sink=sink, overlap=C5["overlap"],
                                        want_roi=True, num_iter=C5["num_iter"], search_channels=(0,),
                                        on_chunk=lambda out: firsts.append(out["first_timepoint"]))
    ret[rank] = (table.cpu().numpy().copy(), (lo, hi), firsts, sorted(sink.files))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_c5_series_streamed_by_two_ranks_equals_one_process(tmp_path):
    """SURVEY 8e, C5's partition: ONE tiled OME-BigTIFF series, two ranks (sharing the box's GPU
    over gloo) stream the two halves of its time axis -- reader.iter_time_chunks(time_range=) ->
    stack.process_stream(first_timepoint=lo) -> SaveSink -> marker-table all-gather (distributed.stream_series).
    The union of what the ranks saved equals the single-process stream FILE BY FILE (same names, same variables, same
    values: assay indices and seeds are global), and every rank ends with the whole series' marker table."""
    import torch.multiprocessing as mp

    import magnify_amd as mg
    from magnify_amd import distributed as mgd
    from magnify_amd import hotpath

    hotpath.require_gpu()
    files = tmp_path / "series"
    files.mkdir()
    pattern = _c5_series(str(files))
    one_dir, two_dir = tmp_path / "one", tmp_path / "two"
    one_dir.mkdir()
    two_dir.mkdir()
    sink = mg.SaveSink(str(one_dir / "t{index:03d}.nc"))
    want_table, span = mgd.stream_series(pattern, C5["chunk"], _c5_flat(), 90.0, seed=9, sink=sink, overlap=C5["overlap"],
                                         want_roi=True, num_iter=C5["num_iter"], search_channels=(0,))
    want_table = want_table.cpu().numpy()
    assert span == (0, C5["n_t"]) and sorted(sink.files) == list(range(C5["n_t"]))
    assert len(want_table) >= 5 * C5["n_t"], "the synthetic series holds too few beads for the comparison to mean much"
    world, port = 2, _free_port()
    manager = mp.get_context("spawn").Manager()
    ret = manager.dict()
    mp.spawn(_c5_worker, args=(world, port, pattern, str(two_dir), ret), nprocs=world, join=True)
    half = C5["n_t"] // 2
    for rank in range(world):
        table, (lo, hi), firsts, saved = ret[rank]
        assert (lo, hi) == (rank * half, (rank + 1) * half)
        assert firsts == list(range(lo, hi, C5["chunk"])) and saved == list(range(lo, hi))
        np.testing.assert_array_equal(table, want_table)  # gathered: the whole series, in time order, on every rank
    assert sorted(p.name for p in two_dir.iterdir()) == sorted(p.name for p in one_dir.iterdir()) == \
        [f"t{t:03d}.nc" for t in range(C5["n_t"])]
    for t in range(C5["n_t"]):
        a, b = mg.load(one_dir / f"t{t:03d}.nc"), mg.load(two_dir / f"t{t:03d}.nc")
        assert set(a.variables) == set(b.variables) and {"roi", "fg", "bg", "x", "y", "fg_sum", "bg_sum"} <= set(a.variables)
        for name in a.variables:
            np.testing.assert_array_equal(np.asarray(a.variables[name].values), np.asarray(b.variables[name].values),
                                          err_msg=f"timepoint {t}: {name}")
        rows = want_table[want_table[:, 0] == t]
        np.testing.assert_array_equal(np.asarray(a["fg_sum"].values)[:, :, 0], rows[:, 6:8])  # the table IS the files' reductions
        np.testing.assert_array_equal(np.asarray(a.coords["y"].values)[:, 0], rows[:, 1])


def test_c5_stream_bench_splits_over_ranks(tmp_path):
    """tools/c5_stream_bench.py --gpus 2 as typed: the parent starts the ranks (it never touches the GPU), rank 0 writes
    the files, both stream their half, the line reports the slowest rank -- and the markers of the single-rank run."""
    import json
    import subprocess

    def run(*flags):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "c5_stream_bench.py"), "--timepoints", "4", "--chunk", "2",
               "--channels", "2", "--grid", "2", "--tile", "200", "--overlap", "20", "--num-iter", "40000", *flags]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert done.returncode == 0, done.stderr[-3000:]
        lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, done.stdout[-2000:]
        return json.loads(lines[0])

    one = run("--files", str(tmp_path / "f1"), "--sink", "save", "--want-roi", "--reader-only")
    two = run("--files", str(tmp_path / "f2"), "--sink", "save", "--want-roi", "--gpus", "2")
    mem = run("--gpus", "2")
    assert one["ranks"]["world_size"] == 1 and one["reader_alone"]["GBs"] > 0
    assert two["ranks"] == {"world_size": 2, "backend": "gloo", "shared_gpu": True, "timepoints_per_rank": 2}
    assert one["markers"] == two["markers"] == mem["markers"] > 0
    assert two["roi_pixels_to_sink"] and two["ms_per_timepoint"] > 0
