"""Stage-wise parity of the HIP kernels (through the C ABI) against the CPU oracle.

Bit-exact for every integer/byte/index stage.  Floating point: the float32 gradient angle is
compared within 2 ulp (the reference's np.arctan2(float32) is platform-dependent in its last
bits), scores within 1 float32 ulp when fed identical angles (perimeter angle table: libm vs
NumPy-SIMD atan2 differ in the last float64 bit).
"""
import numpy as np
import pytest

from oracle import ref_numeric as rn
from oracle import ref_opencv as rcv
from oracle import ref_pipeline as rp
from synth import draw_beads, noisy_bead_image, vignette

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hp():
    from magnify_amd import hotpath

    hotpath.require_gpu()  # fails loudly if the HIP library or the GPU is missing
    return hotpath


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ulp_diff_f32(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


# ---------------------------------------------------------------------------------------------
# A1 / A2
# ---------------------------------------------------------------------------------------------


@pytest.mark.parametrize("overlap", [0, 5, 8])
@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float64, np.uint8])
def test_stitch_copy(hp, overlap, dtype):
    rng = np.random.default_rng(1)
    shape = (2, 3, 2, 3, 40, 48)
    tiles = (rng.random(shape) * 200).astype(dtype)
    img, _ = hp.flatfield_stitch(dev(tiles), overlap, apply_flatfield=False, want_minmax=False)
    np.testing.assert_array_equal(img.cpu().numpy(), rp.stitch(tiles, overlap))


def test_stitch_reference_assertions(hp):
    # tests/test_stitch.py:9-26 (basic), :28-45 (single tile), :78-96 (zero overlap)
    rng = np.random.default_rng(2)
    t = rng.random((1, 1, 2, 3, 40, 40))
    img = hp.flatfield_stitch(dev(t), 5, apply_flatfield=False, want_minmax=False)[0].cpu().numpy()
    assert img.shape[-2:] == (2 * 35, 3 * 35)
    np.testing.assert_array_equal(img[0, 0, 35:70, 35:70], t[0, 0, 1, 1, 2:37, 2:37])
    t = rng.random((1, 1, 1, 1, 30, 30))
    img = hp.flatfield_stitch(dev(t), 5, apply_flatfield=False, want_minmax=False)[0].cpu().numpy()
    np.testing.assert_array_equal(img[0, 0], t[0, 0, 0, 0, 2:27, 2:27])
    t = rng.random((1, 1, 1, 2, 20, 20))
    img = hp.flatfield_stitch(dev(t), 0, apply_flatfield=False, want_minmax=False)[0].cpu().numpy()
    np.testing.assert_array_equal(img[0, 0, :, :20], t[0, 0, 0, 0])
    np.testing.assert_array_equal(img[0, 0, :, 20:], t[0, 0, 0, 1])
    with pytest.raises(ValueError):
        hp.flatfield_stitch(dev(t), -5)
    with pytest.raises(ValueError):
        hp.flatfield_stitch(dev(rng.random((1, 1, 2, 2, 50, 50))), 100)


@pytest.mark.parametrize("case", ["scalar", "image", "image64", "default"])
def test_flatfield_u16(hp, case):
    rng = np.random.default_rng(3)
    tiles = rng.integers(90, 4000, size=(2, 2, 2, 2, 64, 72), dtype=np.uint16)
    if case == "scalar":
        flat, dark = 0.8, 100.0
    elif case == "image":
        flat, dark = vignette((64, 72)), 100.0
    elif case == "image64":
        flat, dark = vignette((64, 72), dtype=np.float64), rng.integers(80, 120, size=(64, 72)).astype(np.uint16)
    else:
        flat, dark = 1.0, 0.0
    for overlap in (0, 7):
        img, minmax = hp.flatfield_stitch(dev(tiles), overlap, flat, dark)
        want = rp.stitch(rp.flatfield_correct(tiles, flat, dark), overlap)
        got = img.cpu().numpy()
        np.testing.assert_array_equal(got, want)
        mm = minmax.cpu().numpy().reshape(2, 2, 2)
        np.testing.assert_array_equal(mm[..., 0], want.min(axis=(-1, -2)))
        np.testing.assert_array_equal(mm[..., 1], want.max(axis=(-1, -2)))


def test_flatfield_f32_and_ragged(hp):
    rng = np.random.default_rng(4)
    tiles = rng.normal(1000, 200, size=(1, 2, 1, 1, 33, 47)).astype(np.float32)
    img, _ = hp.flatfield_stitch(dev(tiles), 0, vignette((33, 47)), 50.0)
    np.testing.assert_array_equal(img.cpu().numpy(), rp.stitch(rp.flatfield_correct(tiles, vignette((33, 47)), 50.0), 0))
    t16 = rng.integers(0, 65536, size=(1, 1, 1, 1, 31, 45), dtype=np.uint16)  # odd sizes: unaligned rows
    img, _ = hp.flatfield_stitch(dev(t16), 3, 0.9, 10.0)
    np.testing.assert_array_equal(img.cpu().numpy(), rp.stitch(rp.flatfield_correct(t16, 0.9, 10.0), 3))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float64])
def test_flatfield_reference_defaults(hp, dtype):
    """flat 1.0 / dark 0.0 (preprocess.py:62): the identity on integer pixels (the device only crops and copies: no
    maxima pass, no arithmetic), NOT on floating point ones (x * M / M rounds); all-zero planes; maxima 0."""
    rng = np.random.default_rng(11)
    hi = 255 if dtype == np.uint8 else 60000
    tiles = (rng.random((2, 2, 2, 3, 40, 56)) * hi).astype(dtype)
    tiles[1, 0] = 0  # an all-zero channel / time inside a non-zero array
    assert hp.flatfield_is_identity(torch.from_numpy(tiles).dtype, 1.0, 0.0) == (np.dtype(dtype).kind == "u")
    assert not hp.flatfield_is_identity(torch.uint16, 1.0, 1.0) and not hp.flatfield_is_identity(torch.uint16, vignette((40, 56)), 0.0)
    for overlap in (0, 6):
        img, minmax = hp.flatfield_stitch(dev(tiles), overlap)
        want = rp.stitch(rp.flatfield_correct(tiles, 1.0, 0.0), overlap)
        np.testing.assert_array_equal(img.cpu().numpy(), want)
        mm = minmax.cpu().numpy().reshape(2, 2, 2)
        np.testing.assert_array_equal(mm[..., 0], want.min(axis=(-1, -2)))
        np.testing.assert_array_equal(mm[..., 1], want.max(axis=(-1, -2)))
    zeros = np.zeros((1, 1, 1, 1, 32, 48), dtype=dtype)
    if np.dtype(dtype).kind == "u":  # 0 * 0 / 0: NaN -> 0 in the integer cast, here and in the oracle
        img, _ = hp.flatfield_stitch(dev(zeros), 0)
        assert not img.cpu().numpy().any()


# ---------------------------------------------------------------------------------------------
# A3 / A4: to_uint8 + blur
# ---------------------------------------------------------------------------------------------


def _blur_case(hp, planes, passthrough=False):
    p, h, w = planes.shape
    cf = hp.CircleFinder(p, h, w, 5, 8, 10)
    d = dev(planes)
    mm = None if passthrough else hp.plane_minmax(d)
    cf.u8 = torch.empty((p, h, w), dtype=torch.uint8, device="cuda")
    from magnify_amd import _native as nat

    nat.check(nat.lib().mg_to_uint8_blur(d.data_ptr(), nat.dtype_code(d.dtype), p, d.stride(0), h, w, d.stride(1),
                                         0 if mm is None else mm.data_ptr(), cf.blur.data_ptr(), cf.u8.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "blur")
    return cf.u8.cpu().numpy(), cf.blur.cpu().numpy()


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float64, np.uint8])
@pytest.mark.parametrize("shape", [(3, 97, 113), (1, 200, 300), (2, 3, 5), (1, 1, 7), (1, 33, 1), (1, 2, 2)])
def test_to_uint8_blur(hp, dtype, shape):
    rng = np.random.default_rng(5)
    if np.issubdtype(dtype, np.integer):
        planes = rng.integers(0, np.iinfo(dtype).max // 3, size=shape).astype(dtype)
    else:
        planes = rng.normal(500, 100, size=shape).astype(dtype)
    u8, blur = _blur_case(hp, planes)
    for k in range(shape[0]):
        want = rn.to_uint8(planes[k])
        np.testing.assert_array_equal(u8[k], want)
        np.testing.assert_array_equal(blur[k], rcv.gaussian_blur5(want))


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8])
@pytest.mark.parametrize("shape", [(2, 300, 512), (1, 257, 248), (3, 70, 252), (1, 2, 8), (1, 700, 1000), (2, 64, 260), (1, 5, 12)])
def test_blur_and_histogram_in_one_pass(hp, dtype, shape):
    """mg_to_uint8_blur_hist (one kernel: strips of 248 columns x 64 rows per wave, neighbours by wave shifts) against the
    oracle's blur and against the two-pass histogram (mg_scharr_hist mode 0) of the same planes."""
    from magnify_amd import _native as nat

    rng = np.random.default_rng(shape[1] * 7 + shape[2])
    p, h, w = shape
    planes = rng.integers(0, np.iinfo(dtype).max // 3, size=shape).astype(dtype)
    planes[0, :, : w // 2] //= 16  # a flat-ish half: many zero magnitudes
    d = dev(planes)
    mm = hp.plane_minmax(d)
    lib, s = nat.lib(), torch.cuda.current_stream().cuda_stream
    words = int(lib.mg_blur_hist_scratch_words(p, h, w))
    assert words >= int(lib.mg_scharr_hist_scratch_words(p, h, w, 0))
    scratch = torch.empty((words,), dtype=torch.int32, device="cuda")
    blur = torch.zeros((p, h, w), dtype=torch.uint8, device="cuda")
    hist = torch.zeros((p, 12288), dtype=torch.int32, device="cuda")
    nat.check(lib.mg_to_uint8_blur_hist(d.data_ptr(), nat.dtype_code(d.dtype), p, d.stride(0), h, w, d.stride(1), mm.data_ptr(),
                                        blur.data_ptr(), 0, hist.data_ptr(), scratch.data_ptr(), words, s), "blur_hist")
    for k in range(p):
        np.testing.assert_array_equal(blur[k].cpu().numpy(), rcv.gaussian_blur5(rn.to_uint8(planes[k])))
    hist2 = torch.zeros_like(hist)
    nat.check(lib.mg_scharr_hist(blur.data_ptr(), p, h, w, 0, 0, hist2.data_ptr(), scratch.data_ptr(), words, s), "hist")
    assert torch.equal(hist, hist2)
    assert hist.sum(dim=1).tolist() == [h * w] * p
    # a view with padded rows and planes (strides that keep the 4-element alignment), and the un-blurred copy asked for
    # (the two-pass route behind the same entry point)
    wide = torch.zeros((p, h + 3, w + 8), dtype=d.dtype, device="cuda")
    wide[:, :h, :w] = d
    view = wide[:, :h, :w]
    blur3, hist3, u8 = torch.zeros_like(blur), torch.zeros_like(hist), torch.zeros_like(blur)
    nat.check(lib.mg_to_uint8_blur_hist(view.data_ptr(), nat.dtype_code(d.dtype), p, view.stride(0), h, w, view.stride(1),
                                        mm.data_ptr(), blur3.data_ptr(), 0, hist3.data_ptr(), scratch.data_ptr(), words, s), "blur_hist")
    assert torch.equal(blur3, blur) and torch.equal(hist3, hist)
    blur3.zero_(), hist3.zero_()
    nat.check(lib.mg_to_uint8_blur_hist(d.data_ptr(), nat.dtype_code(d.dtype), p, d.stride(0), h, w, d.stride(1), mm.data_ptr(),
                                        blur3.data_ptr(), u8.data_ptr(), hist3.data_ptr(), scratch.data_ptr(), words, s), "blur_hist")
    assert torch.equal(blur3, blur) and torch.equal(hist3, hist)
    np.testing.assert_array_equal(u8[0].cpu().numpy(), rn.to_uint8(planes[0]))


def test_to_uint8_golden_and_constant(hp, golden):
    g = golden("to_uint8")
    for key in ("a16", "a16n", "af32", "af64", "const"):
        u8, _ = _blur_case(hp, g[key][None])
        np.testing.assert_array_equal(u8[0], g[key + "_out"])
    planes = np.random.default_rng(6).integers(0, 256, size=(2, 40, 50), dtype=np.uint8)
    u8, blur = _blur_case(hp, planes, passthrough=True)
    np.testing.assert_array_equal(u8, planes)
    np.testing.assert_array_equal(blur[1], rcv.gaussian_blur5(planes[1]))


# ---------------------------------------------------------------------------------------------
# A5-A7: edges
# ---------------------------------------------------------------------------------------------


def _edge_images():
    imgs = [noisy_bead_image(7, (300, 420), 12, r_lo=6, r_hi=14)[0],
            draw_beads((300, 420), [[60, 60], [150, 200], [250, 380], [5, 300]], 20),
            noisy_bead_image(8, (300, 420), 6, poisson=400.0, read_noise=30.0)[0]]
    return np.stack(imgs)


@pytest.mark.parametrize("quantiles", [(0.1, 0.9), (0.5, 0.99), (0.9, 0.1)])
def test_edge_stage(hp, quantiles):
    planes = _edge_images()
    p, h, w = planes.shape
    cf = hp.CircleFinder(p, h, w, 5, 14, 1000)
    cf.keep_debug_maps = True
    n_edges = cf.edge_stage(dev(planes), None, *quantiles)
    bits = cf.edge_bits.cpu().numpy().view(np.uint32)
    edges = cf.edges.cpu().numpy()
    angle = cf.angle.cpu().numpy()
    counts, starts = cf.cell_counts.cpu().numpy(), cf.cell_starts.cpu().numpy()
    coords = cf.coords.cpu().numpy()
    for k in range(p):
        u8 = rn.to_uint8(planes[k])
        blur, dx, dy, want_edges, (lo, hi) = rp.edge_stage(u8, *quantiles)
        np.testing.assert_array_equal(cf.blur[k].cpu().numpy(), blur)
        assert (np.float32(lo), np.float32(hi)) == tuple(cf.quantiles[k])
        assert tuple(cf.thresh[k].cpu().numpy()) == rcv.canny_thresholds(lo, hi)
        np.testing.assert_array_equal(edges[k], want_edges)
        assert n_edges[k] == want_edges.sum()
        packed = np.unpackbits(bits[k].view(np.uint8), bitorder="little")[: h * w].reshape(h, w)
        np.testing.assert_array_equal(packed, want_edges)
        assert not np.unpackbits(bits[k].view(np.uint8), bitorder="little")[h * w:].any()
        # the weak bitmap is OpenCV's NMS map != 1 (candidates), the strong seed set its == 2
        nms_map = rcv.canny_nms(dx.astype(np.int16), dy.astype(np.int16), *rcv.canny_thresholds(lo, hi))
        weak = np.unpackbits(cf.weak_bits[k].cpu().numpy().view(np.uint8), bitorder="little")[: h * w].reshape(h, w)
        np.testing.assert_array_equal(weak, (nms_map != 1).astype(np.uint8))
        # orientation class planes (for the scoring prefilter): floor((atan2(dy, dx) mod pi) / (pi / 4)),
        # decided on the integer gradient; checked where the class is unambiguous (off the boundaries)
        cb = cf.class_bits[k].cpu().numpy().view(np.uint32)
        c0 = np.unpackbits(cb[0].view(np.uint8), bitorder="little")[: h * w].reshape(h, w)
        c1 = np.unpackbits(cb[1].view(np.uint8), bitorder="little")[: h * w].reshape(h, w)
        idx, idy = dx.astype(np.int64), dy.astype(np.int64)
        phi = np.arctan2(idy.astype(np.float64), idx.astype(np.float64)) % np.pi
        interior = (idx != 0) & (idy != 0) & (np.abs(idx) != np.abs(idy))
        want_class = np.floor(phi / (np.pi / 4)).astype(np.int64)
        np.testing.assert_array_equal((2 * c1 + c0)[interior], want_class[interior])
        # on the boundaries either neighbouring class is acceptable
        got = (2 * c1 + c0).astype(np.int64)
        on_b = ~interior & ((idx != 0) | (idy != 0))
        near = np.round(phi / (np.pi / 4)).astype(np.int64) % 4  # the boundary index k: classes k-1 and k
        assert np.all((got[on_b] == near[on_b]) | (got[on_b] == (near[on_b] - 1) % 4))
        # third plane: the half of the quarter -> bins of pi/8 (never ambiguous off the quarter boundaries:
        # tan(pi/8), tan(3 pi/8) are irrational), bin = 2 * quarter + c2
        c2 = np.unpackbits(cb[2].view(np.uint8), bitorder="little")[: h * w].reshape(h, w)
        want_bin = np.floor(phi / (np.pi / 8)).astype(np.int64)
        np.testing.assert_array_equal((4 * c1 + 2 * c0 + c2)[interior], want_bin[interior])
        got_bin = (4 * c1 + 2 * c0 + c2).astype(np.int64)
        near8 = np.round(phi / (np.pi / 8)).astype(np.int64) % 8
        assert np.all((got_bin[on_b] == near8[on_b]) | (got_bin[on_b] == (near8[on_b] - 1) % 8))
        # angle map: sentinel off-edge; on edges the correctly rounded float32 arctan2, which is
        # within 2 ulp of NumPy's SIMD float32 arctan2 (itself not correctly rounded: e.g.
        # arctan2(-1, 1) comes out 1 ulp above float32(-pi/4) on AVX-512 hosts)
        on = want_edges > 0
        want_angle = np.arctan2(dy, dx)
        assert ulp_diff_f32(angle[k][on], want_angle[on]).max() <= 2
        exact64 = np.arctan2(dy.astype(np.float64), dx.astype(np.float64)).astype(np.float32)
        assert (angle[k][on] == exact64[on]).mean() > 0.9999
        gcoords, gstarts, gcounts = rn.grid_array(want_edges, 20)
        np.testing.assert_array_equal(counts[k].reshape(gcounts.shape), gcounts)
        np.testing.assert_array_equal(starts[k].reshape(gstarts.shape), gstarts)
        np.testing.assert_array_equal(coords[k, : len(gcoords)], gcoords)


def test_edge_stage_empty_and_tiny(hp):
    planes = np.zeros((2, 64, 64), dtype=np.uint16)
    planes[1, 30:34, 30:34] = 500
    cf = hp.CircleFinder(2, 64, 64, 5, 8, 100)
    cf.keep_debug_maps = True
    n_edges = cf.edge_stage(dev(planes), None, 0.1, 0.9)
    for k in range(2):
        _, _, _, want, _ = rp.edge_stage(rn.to_uint8(planes[k]), 0.1, 0.9)
        np.testing.assert_array_equal(cf.edges[k].cpu().numpy(), want)
    assert n_edges[0] == 0


@pytest.mark.parametrize("quantiles", [(0.1, 0.9), (0.45, 0.55), (0.999, 0.2)])
def test_edge_thresholds_window_passes(hp, quantiles):
    """Quantile ranks inside COARSE histogram bins (strong gradients everywhere: white noise over the full range, a
    noiseless plane searched near its top quantile) are resolved by window passes that stay on the device; planes of
    one batch need different numbers of them (0 .. 4).  Quantiles, thresholds and edges equal the oracle's, and the
    second call on the finder runs the passes optimistically (one host round trip)."""
    rng = np.random.default_rng(11)
    shape = (200, 264)
    planes = np.stack([rng.integers(0, 65536, shape).astype(np.uint16),                    # ranks in several coarse bins
                       noisy_bead_image(21, shape, 5, r_lo=6, r_hi=12)[0],                 # fine bins only
                       draw_beads(shape, [[50, 60], [120, 200], [160, 90]], 24),           # noiseless
                       (rng.integers(0, 2, shape) * 40000 + 500).astype(np.uint16)])       # two-level noise
    p, h, w = planes.shape
    cf = hp.CircleFinder(p, h, w, 5, 13, 2000)
    cf.keep_debug_maps = True
    for call in range(2):
        got, _ = cf.find(dev(planes), None, quantiles[0], quantiles[1], 0.3, 5, [3, 4, 5, 6])
        assert cf.stats["optimistic"] == (call == 1)
        for k in range(p):
            _, _, _, want_edges, (lo, hi) = rp.edge_stage(rn.to_uint8(planes[k]), *quantiles)
            assert (np.float32(lo), np.float32(hi)) == tuple(cf.quantiles[k]), (call, k)
            assert tuple(cf.thresh[k].cpu().numpy()) == rcv.canny_thresholds(lo, hi), (call, k)
            np.testing.assert_array_equal(cf.edges[k].cpu().numpy(), want_edges)
    assert cf.stats["hist_passes"] > 1  # the white-noise plane cannot do without window passes


def test_hysteresis_reaches_the_fixed_point(hp):
    """mg_canny_hysteresis, swept until no tile asks for another turn, against connected components (8-connected weak
    regions that hold a strong pixel, scipy.ndimage.label): a serpentine weak line with ONE strong seed that crosses
    the 256 x 256 tile borders dozens of times in both directions (growth is handed from tile to tile and back through
    the neighbour flags), random weak clutter with sparse seeds, lines along tile borders and through tile corners, an
    empty plane.  With the tile flags (only tiles a neighbour asked for are worked on) and without (every tile, every
    sweep): same bitmaps, same number of sweeps."""
    from scipy import ndimage

    from magnify_amd import _native as nat

    rng = np.random.default_rng(5)
    h, w = 700, 900
    weak = np.zeros((4, h, w), dtype=bool)
    strong = np.zeros((4, h, w), dtype=bool)
    # plane 0: serpentine, 12 px pitch, columns 5 .. w - 6, every row pair connected at alternating ends
    for j, y in enumerate(range(4, h - 13, 12)):
        weak[0, y, 5: w - 5] = True
        weak[0, y: y + 13, (w - 6) if j % 2 == 0 else 5] = True
    strong[0, 4, 5] = True
    # plane 1: clutter
    weak[1] = rng.random((h, w)) < 0.35
    strong[1] = weak[1] & (rng.random((h, w)) < 0.002)
    # plane 2: lines on the tile borders (rows 255 / 256, cols 255 / 256, 511 / 512), diagonal steps through the corners
    weak[2, 255, :] = weak[2, :, 256] = weak[2, 512, :] = weak[2, :, 511] = True
    for d in range(-20, 21):
        weak[2, 256 + d, 256 - d] = True
    strong[2, 255, 0] = True
    strong[2, 699, 511] = True
    # plane 3: nothing
    want = np.zeros_like(weak)
    for p in range(4):
        lab, _ = ndimage.label(weak[p], structure=np.ones((3, 3), dtype=int))
        keep = np.unique(lab[strong[p] & weak[p]])
        want[p] = np.isin(lab, keep[keep > 0]) | strong[p]
    words = 2 * ((h * w + 63) // 64) + 2

    def pack(m):
        out = np.zeros((4, words), dtype=np.uint32)
        for p in range(4):
            b = np.packbits(m[p].reshape(-1), bitorder="little")
            out[p].view(np.uint8)[: len(b)] = b
        return dev(out.view(np.int32))

    def unpack(t):
        a = t.cpu().numpy().view(np.uint32)
        return np.stack([np.unpackbits(a[p].view(np.uint8), bitorder="little")[: h * w].reshape(h, w).astype(bool) for p in range(4)])

    d_weak = pack(weak)
    tx, ty = nat.C.c_int(0), nat.C.c_int(0)
    nat.check(nat.lib().mg_hysteresis_tiles(h, w, nat.C.byref(tx), nat.C.byref(ty)), "mg_hysteresis_tiles")
    stream = torch.cuda.current_stream().cuda_stream
    changed = torch.zeros((4,), dtype=torch.int32, device="cuda")
    sweeps = {}
    for use_flags in (False, True):
        d_strong = pack(strong)
        flags = [torch.zeros((4, ty.value, tx.value), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for sweep in range(2000):
            changed.zero_()
            flags[(sweep + 1) % 2].zero_()
            nat.check(nat.lib().mg_canny_hysteresis(d_weak.data_ptr(), d_strong.data_ptr(), words, 4, h, w, changed.data_ptr(),
                                                    flags[sweep % 2].data_ptr() if (use_flags and sweep) else 0,
                                                    flags[(sweep + 1) % 2].data_ptr() if use_flags else 0, stream),
                      "mg_canny_hysteresis")
            if not changed.cpu().numpy().any():
                break
        sweeps[use_flags] = sweep
        got = unpack(d_strong)
        for p in range(4):
            np.testing.assert_array_equal(got[p], want[p], err_msg=f"plane {p}, flags {use_flags}")
    assert 20 < sweeps[True] < 1999 and sweeps[True] == sweeps[False]  # the serpentine really needs many hand-overs


# ---------------------------------------------------------------------------------------------
# A8-A11: candidates, unique circles, scores, suppression
# ---------------------------------------------------------------------------------------------


@pytest.mark.parametrize("path", ["keyed", "keyed_tile_score", "atomic"])
def test_candidates_scores_nms(hp, path):
    """keyed: de-duplication by 32-bit keys and per-tile LDS bitmaps (no global atomics), scoring by
    mg_score_circles_keyed (group-per-circle prefilter with orientation bounds, angles on demand);
    keyed_tile_score: same keys, scored by mg_score_circles (lane-per-circle prefilter, angle map);
    atomic: atomicOr into the global bitmap + mg_score_circles.  Same unique circle list, same scores for
    every circle that can pass, same suppression result on all three."""
    keyed = path != "atomic"
    planes = _edge_images()
    p, h, w = planes.shape
    min_r, max_r, num_iter, min_dist = 5, 14, 20000, 5
    cf = hp.CircleFinder(p, h, w, min_r, max_r, num_iter)
    assert cf.keyed and cf.keyed_score
    cf.keyed = keyed
    cf.keyed_score = path == "keyed"
    cf.keep_debug_maps = True
    seeds = [11, 12, 23]  # (plane 2 is nearly all noise: a seed with which the 20 000 draws hit one of its few beads)
    res, _ = cf.find(dev(planes), None, 0.1, 0.9, 0.3, min_dist, seeds, keep_raw=True)
    raw = cf.raw.cpu().numpy()
    circles = cf.circles.cpu().numpy()
    n_circles = cf.num_circles.cpu().numpy()
    scores = cf.scores.cpu().numpy()
    angle = cf.angle.cpu().numpy()
    if not keyed:
        assert cf.bitmap.count_nonzero().item() == 0  # the atomic path's compaction leaves the bitmap clean
    else:
        # keyed path: unique 32-bit keys, one slice per tile (slices in arrival order, sorted inside);
        # decode them, bring list and scores into the canonical order = ascending key
        ukeys = cf.unique_keys.cpu().numpy().view(np.uint32)
        ranges = cf.tile_ranges.cpu().numpy()
        alive_idx, n_alive = cf.alive.cpu().numpy(), cf.num_alive.cpu().numpy()
        ntc = (w + 2 * max_r + 63) // 64
        dec_circles, dec_scores = [], []
        for k in range(p):
            n = n_circles[k]
            kk = ukeys[k, :n]
            assert ranges[k, :, 1].sum() == n and len(np.unique(kk)) == n
            for t in np.nonzero(ranges[k, :, 1])[0][:50]:
                sl = kk[ranges[k, t, 0]: ranges[k, t, 0] + ranges[k, t, 1]]
                assert (sl >> 17 == t).all() and (np.diff(sl.astype(np.int64)) > 0).all()
            tile = (kk >> 17).astype(np.int64)
            dec = np.stack([(tile // ntc) * 64 + ((kk >> 6) & 63) - max_r, (tile % ntc) * 64 + (kk & 63) - max_r,
                            min_r + ((kk >> 12) & 31)], axis=1).astype(np.int32)
            # the (row, col, r) triples exist in cf.circles only for the circles that passed the threshold
            ai = alive_idx[k, : n_alive[k]]
            np.testing.assert_array_equal(circles[k, ai], dec[ai])
            order = np.argsort(kk, kind="stable")
            dec_circles.append(dec[order])
            dec_scores.append(scores[k, :n][order])
    for k in range(p):
        u8 = rn.to_uint8(planes[k])
        _, dx, dy, edges, _ = rp.edge_stage(u8, 0.1, 0.9)
        picks = rn.draw_picks(seeds[k], num_iter, edges, 20)
        cand = rn.candidate_circles_from_picks(edges, 20, *picks)
        np.testing.assert_array_equal(raw[k].view(np.uint32), cand.view(np.uint32))  # bit-exact incl. NaN/inf
        # step 4 + de-duplication + (r, row, col) order
        with np.errstate(invalid="ignore"):
            c = cand[(cand[:, 2] >= min_r) & (cand[:, 2] <= max_r)]
            c = np.round(c).astype(np.int32)
        c = c[(c[:, 0] + c[:, 2] >= 0) & (c[:, 1] + c[:, 2] >= 0) & (c[:, 0] - c[:, 2] < h) & (c[:, 1] - c[:, 2] < w)]
        c = np.unique(c, axis=0)
        c = c[np.lexsort(rn.canonical_key(c, max_r))]  # tile-major emission order
        assert n_circles[k] == len(c)
        np.testing.assert_array_equal(dec_circles[k] if keyed else circles[k, : len(c)], c)
        # scores, given the GPU's own angle map as the oracle's grad_angles
        ang = np.where(edges > 0, angle[k], 0).astype(np.float32)
        pad = 2 * max_r
        pa, pe = np.pad(ang, pad), np.pad(edges, pad)
        want_scores = np.empty(len(c), dtype=np.float32)
        for r in range(min_r, max_r + 1):
            sel = c[:, 2] == r
            per = rn.circle_points(r)
            want_scores[sel] = rn.mean_grad(pa, pe, c[sel, :2] + pad, per) / len(per)
        got = (dec_scores[k] if keyed else scores[k, : len(c)]).copy()
        # circles the exact prefilter skipped are provably below the threshold
        skipped = got == np.float32(-2.0)
        assert (want_scores[skipped] < np.float32(0.3)).all()
        assert skipped.mean() > 0.2  # the prefilter does remove work
        assert ulp_diff_f32(got[~skipped], want_scores[~skipped]).max() <= 1
        assert (got[~skipped] == want_scores[~skipped]).mean() > 0.999
        got[skipped] = want_scores[skipped]
        # suppression, given the GPU's scores
        good = got >= np.float32(0.3)
        cc, ss = c[good], got[good]
        perm = rn.canonical_order(cc, ss, max_r)
        cc, ss = cc[perm], ss[perm]
        keep = rn.filter_neighbors(cc, min_dist)
        np.testing.assert_array_equal(res[k][0], cc[keep])
        np.testing.assert_array_equal(res[k][1], ss[keep])
        assert len(res[k][0]) >= (4 if k < 2 else 1)
        # and the whole thing against the oracle's find_circles (same RNG stream, GPU angles)
        oc, osc = rp.find_circles(u8, 0.1, 0.9, 20, num_iter, min_r, max_r, 0.3, min_dist, seed=seeds[k],
                                  grad_angles=ang)
        np.testing.assert_array_equal(res[k][0], oc)
        assert ulp_diff_f32(res[k][1], osc).max() <= 1
        # end to end with the oracle's OWN angles (np.arctan2 on float32, utils.py:170 -- NumPy's SIMD
        # loop, up to 2 ulp from the GPU's correctly rounded angles): same circles in the same order;
        # a score may move by a few ulp, which could only reorder / flip circles whose scores are that
        # close to each other or to min_roundness -- none here (bound stated in DESIGN.md section 2)
        oc2, osc2 = rp.find_circles(u8, 0.1, 0.9, 20, num_iter, min_r, max_r, 0.3, min_dist, seed=seeds[k])
        np.testing.assert_array_equal(res[k][0], oc2)
        assert ulp_diff_f32(res[k][1], osc2).max() <= 4


def test_nms_golden_and_wrap(hp, golden):
    """filter_neighbors golden cases (reference output, incl. negative-index wrap) through the
    GPU suppression: feed circles with strictly decreasing scores."""
    from magnify_amd import _native as nat

    g = golden("filter_neighbors")
    for case in range(4):
        c = g[f"circles_{case}"].astype(np.int32)
        min_dist = int(g[f"min_dist_{case}"])
        n = len(c)
        cf = hp.CircleFinder(1, 400, 400, 5, 25, n)
        cf.circles[0, :n] = dev(c)
        cf.num_circles[0] = n
        cf.scores[0, :n] = dev(np.linspace(1.0, 0.5, n).astype(np.float32))
        cf.alive[0, :n] = dev(np.random.default_rng(case).permutation(n).astype(np.int32))
        cf.num_alive[0] = n
        cf.max_rc[0] = dev(np.array([c[:, 0].max(), c[:, 1].max()], dtype=np.int32))
        out, out_scores, num_out = cf.nms_stage(min_dist)
        m = int(num_out[0].item())
        want = c[g[f"valid_{case}"]]
        assert m == len(want)
        np.testing.assert_array_equal(out[0, :m].cpu().numpy(), want)


@pytest.mark.parametrize("min_dist", [3, 8, 15, 16])
def test_nms_decided_from_the_circles_alone(hp, monkeypatch, min_dist):
    """mg_nms_sparse (a plane's alive circles bucketed in LDS, ring conflicts from the bitmap of ring (-) ring) against the
    oracle's filter_neighbors and against the claim-grid rounds: crowded planes with equal scores and shared centres, a
    plane with a centre far outside the image (negative claim index: left to the rounds), an empty plane, a radius
    beyond the kernel's bitmap (16: every plane left to the rounds)."""
    rng = np.random.default_rng(100 + min_dist)
    h, w, P = 700, 900, 4
    sets = []
    for plane in range(P):
        n = [3000, 1200, 0, 2000][plane]
        cl = rng.integers(0, 40, size=(60, 2)) * np.array([17, 22]) + 10  # clusters: many circles within reach of each other
        c = cl[rng.integers(0, 60, n)] + rng.integers(-2 * min_dist, 2 * min_dist + 1, size=(n, 2))
        c = np.column_stack([np.clip(c[:, 0], -min_dist - 1, h), np.clip(c[:, 1], -min_dist - 1, w), rng.integers(5, 20, n)])
        if plane == 1 and n:
            c[7, 0] = -min_dist - 2  # the reference's claim index goes negative: wraps
            c[8] = [h + 20, 33, 9]
        sc = np.round(rng.uniform(0.3, 1.0, n), 2).astype(np.float32)  # two decimals: many equal scores
        if n > 50:
            c[20:40, :2] = c[20, :2]  # one centre, twenty circles
        sets.append((c.astype(np.int32), sc))
    cap = 3200

    def run(sparse):
        monkeypatch.setattr(hp, "_NMS_SPARSE", sparse)
        cf = hp.CircleFinder(P, h, w, 5, 25, cap)
        for plane, (c, sc) in enumerate(sets):
            n = len(c)
            cf.circles[plane, :n] = dev(c)
            cf.scores[plane, :n] = dev(sc)
            cf.alive[plane, :n] = dev(rng.permutation(n).astype(np.int32))
            cf.num_circles[plane] = n
            cf.num_alive[plane] = n
            cf.max_rc[plane] = dev(np.array([c[:, 0].max(), c[:, 1].max()] if n else [0, 0], dtype=np.int32))
        out, out_scores, num_out = cf.nms_stage(min_dist)
        done = cf.nms_done.cpu().numpy().copy() if sparse else None
        assert not cf.state.any()  # the cleanup has returned every state byte to 0
        return [out[k, : int(num_out[k])].cpu().numpy() for k in range(P)], done

    got, done = run(True)
    want, _ = run(False)
    assert done.tolist() == ([0, 0, 0, 0] if min_dist > 15 else [1, 0, 1, 1])
    for plane, (c, sc) in enumerate(sets):
        np.testing.assert_array_equal(got[plane], want[plane], err_msg=f"plane {plane}")
        if len(c):  # (the oracle wraps negative claim indices as numba does)
            order = np.lexsort((np.arange(len(c)), -sc.astype(np.float64)))
            keep = rn.filter_neighbors(c[order], min_dist)
            np.testing.assert_array_equal(got[plane], c[order][keep], err_msg=f"plane {plane} vs oracle")


def test_find_circles_empty(hp):
    planes = np.zeros((1, 128, 128), dtype=np.uint16)
    cf = hp.CircleFinder(1, 128, 128, 8, 12, 1000)
    res, _ = cf.find(dev(planes), None, 0.1, 0.9, 0.3, 8, [5])
    assert res[0][0].shape == (0, 3)


def test_find_circles_no_suppression(hp):
    """min_dist == 0 (per-chamber refinement, find.py:352): all scored circles, priority order."""
    img = draw_beads((72, 72), [[36, 36]], 20)
    cf = hp.CircleFinder(1, 72, 72, 4, 15, 400)
    cf.keep_debug_maps = True
    res, _ = cf.find(dev(img[None]), None, 0.1, 0.99, 0.2, 0, [9])
    u8 = rn.to_uint8(img)
    ang = np.where(cf.edges[0].cpu().numpy() > 0, cf.angle[0].cpu().numpy(), 0).astype(np.float32)
    oc, osc = rp.find_circles(u8, 0.1, 0.99, 20, 400, 4, 15, 0.2, 0, seed=9, grad_angles=ang)
    np.testing.assert_array_equal(res[0][0], oc)
    assert len(oc) > 0 and abs(int(oc[0][0]) - 36) <= 1 and abs(int(oc[0][2]) - 10) <= 1


# ---------------------------------------------------------------------------------------------
# A12-A15, A18: labels, ROI, reductions
# ---------------------------------------------------------------------------------------------


def test_circle_labels(hp, golden):
    g = golden("circle_labels")
    h, w = (int(v) for v in g["shape"])
    beads = g["beads"]
    ok = beads[:, 2] >= 2
    lab = hp.circle_labels([beads[ok]], h, w)[0].cpu().numpy()
    np.testing.assert_array_equal(lab, rn.circle_labels(beads[ok], h, w))
    if ok.all():
        np.testing.assert_array_equal(lab, g["labels"])
    # two assays at once, one empty
    labs = hp.circle_labels([beads[ok][:5], np.empty((0, 3), int)], h, w).cpu().numpy()
    np.testing.assert_array_equal(labs[0], rn.circle_labels(beads[ok][:5], h, w))
    assert (labs[1] == -1).all()


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_roi_gather_reduce(hp, dtype):
    rng = np.random.default_rng(21)
    c, t, h, w, L = 3, 2, 160, 200, 40
    image = rng.integers(0, 4000, size=(c, t, h, w)).astype(dtype)
    beads = np.array([[20, 20, 8], [5, 190, 6], [150, 100, 10], [80, 80, 9], [84, 92, 9], [159, 0, 5], [33, 57, 6],
                      [121, 143, 7]])  # odd and even window offsets
    labels = hp.circle_labels([beads], h, w)
    res = hp.roi_gather_reduce(dev(image)[None], [beads], L, labels)
    lab = rn.circle_labels(beads, h, w)
    m = len(beads)
    roi = np.zeros((m, c, t, L, L), dtype=dtype)
    fg = np.zeros((m, t, L, L), dtype=bool)
    bg = np.zeros_like(fg)
    for i, (row, col, _) in enumerate(beads):
        top, bottom, left, right = rn.bounding_box(int(col), int(row), L, w, h)
        roi[i] = image[:, :, top:bottom, left:right]
        fg[i] = (lab[top:bottom, left:right] == i)[None]
        bg[i] = (lab[top:bottom, left:right] == -1)[None]
    np.testing.assert_array_equal(res["roi"].cpu().numpy(), roi)
    np.testing.assert_array_equal(res["fg"].cpu().numpy().astype(bool), fg[:, 0])
    np.testing.assert_array_equal(res["bg"].cpu().numpy().astype(bool), bg[:, 0])
    red = rp.roi_reduce(roi, fg, bg)
    counts = res["counts"].cpu().numpy()
    np.testing.assert_array_equal(counts[:, 0], red["fg_count"][:, 0])
    np.testing.assert_array_equal(counts[:, 1], red["bg_count"][:, 0])
    sums = res["sums"].cpu().numpy()
    if dtype == np.uint16:
        np.testing.assert_array_equal(sums[..., 0], red["fg_sum"])  # exact integer sums
        np.testing.assert_array_equal(sums[..., 1], red["bg_sum"])
        med = hp.masked_median_u16(res["roi"], res["fg"]).cpu().numpy()
        np.testing.assert_array_equal(med, red["fg_median"])
        med = hp.masked_median_u16(res["roi"], res["bg"]).cpu().numpy()
        np.testing.assert_array_equal(med, red["bg_median"])
    else:
        np.testing.assert_allclose(sums[..., 0], red["fg_sum"], rtol=1e-12)  # float64 sums, other order
        np.testing.assert_allclose(sums[..., 1], red["bg_sum"], rtol=1e-12)


@pytest.mark.parametrize("dtype,L", [(np.uint16, 40), (np.uint16, 100), (np.uint16, 33), (np.float32, 40), (np.uint8, 64)])
def test_roi_segment_reduce_from_disks(hp, dtype, L):
    """Masks straight from the bead table (no label map) equal the masks read from the oracle's
    circle_labels map: overlapping (contested) disks, disks cut by the image border, centres outside
    the image, crowded windows, several assays with different bead counts."""
    rng = np.random.default_rng(33)
    c, t, h, w = 2, 2, 180, 220
    images = rng.integers(0, 250, size=(3, c, t, h, w)).astype(dtype)
    crowded = np.column_stack([rng.integers(40, 140, 150), rng.integers(40, 180, 150), rng.integers(2, 12, 150)])  # > 96 disks per window: overflow path
    assays = [
        np.array([[20, 20, 8], [5, 190, 6], [150, 100, 10], [80, 80, 9], [84, 92, 9], [179, 0, 5], [33, 57, 6],
                  [121, 143, 7], [-3, 100, 6], [90, 224, 7], [80, 84, 25]]),
        crowded,
        np.empty((0, 3), dtype=np.int64),
    ]
    res = hp.roi_gather_reduce(dev(images), assays, L, None, disks=True)
    off = res["offsets"]
    assert off[-1] == sum(len(b) for b in assays)
    for a, beads in enumerate(assays):
        if len(beads) == 0:
            continue
        lab = rn.circle_labels(beads, h, w)
        assert (lab == -2).any()
        for i, (row, col, _) in enumerate(beads):
            g = off[a] + i
            top, bottom, left, right = rn.bounding_box(int(col), int(row), L, w, h)
            sub = lab[top:bottom, left:right]
            np.testing.assert_array_equal(res["fg"][g].cpu().numpy().astype(bool), sub == i)
            np.testing.assert_array_equal(res["bg"][g].cpu().numpy().astype(bool), sub == -1)
            win = images[a][:, :, top:bottom, left:right]
            np.testing.assert_array_equal(res["roi"][g].cpu().numpy(), win)
            sums = res["sums"][g].cpu().numpy()
            np.testing.assert_array_equal(sums[..., 0], win.astype(np.float64)[:, :, sub == i].sum(-1))
            np.testing.assert_array_equal(sums[..., 1], win.astype(np.float64)[:, :, sub == -1].sum(-1))
            assert tuple(res["counts"][g].cpu().numpy()) == (int((sub == i).sum()), int((sub == -1).sum()))
    # the same pass queued before the host knows the counts: padded device tables + device counts + a capacity, the
    # assays' offsets computed on the device (mg_counts_to_offsets); a count above the capacity is cut to it
    if dtype == np.uint16:
        cap = 160
        tab = np.zeros((3, cap + 5, 3), dtype=np.int32)
        for a, beads in enumerate(assays):
            tab[a, : len(beads)] = beads
        d_counts = dev(np.array([len(b) for b in assays], dtype=np.int32))
        late = hp.roi_gather_reduce(dev(images), None, L, None, disks=True, device_tables=(dev(tab), None, 25),
                                    device_counts=(d_counts, cap, None))
        late = hp.finish_roi(late, [len(b) for b in assays])
        np.testing.assert_array_equal(late["offsets"], off)
        for key in ("roi", "fg", "bg", "sums", "counts"):
            np.testing.assert_array_equal(late[key].cpu().numpy(), res[key].cpu().numpy(), err_msg=key)
        cut = hp.roi_gather_reduce(dev(images), None, L, None, disks=True, device_tables=(dev(tab), None, 25),
                                   device_counts=(d_counts, 100, 120))
        cut = hp.finish_roi(cut, [11, 100, 0])
        np.testing.assert_array_equal(cut["roi"][:11].cpu().numpy(), res["roi"][:11].cpu().numpy())
        assert cut["roi"].shape[0] == 111
        assert hp.finish_roi(dict(cut, bound=110), [11, 100, 0]) is None  # more markers than the launch was sized for


# ---------------------------------------------------------------------------------------------
# the optimistic chain (one host round trip per call) against the checked chain (three)
# ---------------------------------------------------------------------------------------------


def test_optimistic_chain_equals_checked_chain(hp):
    """A sequence of calls on one finder: the first goes through the checked chain, the later ones run
    optimistically with the sweeps / rounds / capacities of the calls before -- including calls where those do NOT
    suffice (a coordinate list / output that is too small, too few sweeps / rounds, many more edges and beads than
    before, a blank plane, a noiseless plane) and the chain has to repair itself.  Every call must return exactly what
    a fresh, checked finder returns."""
    shape = (320, 384)

    def normal(k):
        return np.stack([noisy_bead_image(k + j, shape, 6)[0] for j in range(2)])

    def shrink_coords(cf):
        cf.coords = cf.coords[:, :50].contiguous()

    def shrink_output(cf):
        cf._out_cap, cf._out_sets = 1, [None, None]

    def few_rounds(cf):
        cf._recent_rounds[:] = [0]

    def few_sweeps(cf):
        cf._recent_sweeps[:] = [0]

    calls = [  # (planes, what is done to the finder before the call, optimistic expected: True / False / None = either)
        (normal(50), None, False),   # first call: checked
        (normal(60), None, True),
        (normal(70), None, True),
        (normal(80), None, True),
        (normal(90), shrink_coords, False),
        (normal(100), shrink_output, True),
        (normal(110), few_rounds, True),
        (normal(120), few_sweeps, None),
        (np.stack([noisy_bead_image(130 + k, shape, 40, r_lo=6, r_hi=10, poisson=200.0)[0] for k in range(2)]), None, None),
        (np.stack([np.zeros(shape, dtype=np.uint16), noisy_bead_image(141, shape, 6)[0]]), None, None),
        (np.stack([draw_beads(shape, [[100, 100], [200, 250]], 30), noisy_bead_image(151, shape, 6)[0]]), None, None),
        (normal(160), None, None),
        (normal(170), None, True),
    ]
    cf = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    took_fast = []
    for n, (planes, sabotage, expect) in enumerate(calls):
        if n > 0:  # enough sweeps / rounds / list capacity for any of these scenes: only the sabotage decides the path
            cf._recent_sweeps[:], cf._recent_rounds[:] = [12], [12]
            if cf.coords.shape[1] < 80000:  # (the edge counts of these scenes vary between 1 000 and 30 000)
                cf.coords = torch.empty((2, 80000, 2), dtype=torch.int32, device="cuda")
        if sabotage:
            sabotage(cf)
        seeds = [1000 + 2 * n, 1001 + 2 * n]
        got, _ = cf.find(dev(planes), None, 0.1, 0.9, 0.3, 5, seeds)
        took_fast.append(bool(cf.stats["optimistic"]))
        ref = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
        ref.optimistic = False
        want, _ = ref.find(dev(planes), None, 0.1, 0.9, 0.3, 5, seeds)
        assert not ref.stats["optimistic"]
        for p in range(2):
            np.testing.assert_array_equal(got[p][0], want[p][0], err_msg=f"call {n} plane {p}")
            np.testing.assert_array_equal(got[p][1], want[p][1], err_msg=f"call {n} plane {p}")
        np.testing.assert_array_equal(cf.n_edges_host, ref.n_edges_host)
        assert expect is None or took_fast[-1] == expect, (n, took_fast)
    assert len(want[1][0]) >= 4  # the scenes do hold beads


def test_graph_replay_equals_eager_launches(hp):
    """The optimistic chain replayed as one hipGraph launch (same input buffer, same hints: captured when the launch
    sequence first shows, the caller vouching for its buffers) against a finder that launches every kernel eagerly:
    same tables call after call,
    with new images and seeds travelling through the fixed buffers, through a call whose image breaks the hints (many
    more beads: the chain is repaired eagerly) and back, and with the ROI pass queued behind the replay (`follow`)."""
    shape = (320, 384)
    images = [np.stack([noisy_bead_image(300 + 10 * n + j, shape, 6)[0] for j in range(2)]) for n in range(14)]
    images[6] = np.stack([noisy_bead_image(400 + k, shape, 40, r_lo=6, r_hi=10, poisson=200.0)[0] for k in range(2)])
    # two input buffers used in turn: two launch sequences, two graphs -- each is replayed after the OTHER has been
    # captured (memset nodes of the first graph were once seen to run with the second graph's parameters)
    bufs = [torch.empty((2,) + shape, dtype=torch.uint16, device="cuda") for _ in range(2)]
    cf = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    ref = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    ref._graphs = None
    assert cf._graphs is not None
    followed = []
    for n, planes in enumerate(images):
        buf = bufs[n % 2]
        buf.copy_(dev(planes))
        if n > 0:  # generous, steady hints (these small scenes need 2 .. 10 sweeps / rounds): one launch sequence
            cf._recent_sweeps[:], cf._recent_rounds[:] = [12], [12]
            if cf.coords.shape[1] < 80000:
                cf.coords = torch.empty((2, 80000, 2), dtype=torch.int32, device="cuda")
        seeds = [7000 + 2 * n, 7001 + 2 * n]
        got, (d_out, _, d_num) = cf.find(buf, None, 0.1, 0.9, 0.3, 5, seeds, stable_input=True,
                                         follow=lambda out, num, cap: (out.clone(), num.clone()))
        followed.append(cf.follow_result)
        want, _ = ref.find(buf, None, 0.1, 0.9, 0.3, 5, seeds)
        for p in range(2):
            np.testing.assert_array_equal(got[p][0], want[p][0], err_msg=f"call {n} plane {p}")
            np.testing.assert_array_equal(got[p][1], want[p][1], err_msg=f"call {n} plane {p}")
            # what the follow-up saw on the device = the final tables
            k = len(got[p][0])
            assert int(followed[-1][1][p].item()) == k
            np.testing.assert_array_equal(followed[-1][0][p, :k].cpu().numpy(), got[p][0])
        np.testing.assert_array_equal(cf.n_edges_host, ref.n_edges_host)
    assert "graph_error" not in cf.stats, cf.stats.get("graph_error")
    assert cf.graph_replays >= 6 and 2 <= cf.graph_captures <= 10, (cf.graph_replays, cf.graph_captures, cf.calls)
    assert ref.graph_replays == 0
    # a caller that hands in a fresh tensor every call (the chip's gathered chamber windows): small inputs are copied
    # into the finder's own block, the graph is keyed on that; a single plane is always launched eagerly
    st = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    one = hp.CircleFinder(1, shape[0], shape[1], 5, 21, 60000)
    for n, planes in enumerate(images[:6]):
        seeds = [7000 + 2 * n, 7001 + 2 * n]
        if n > 0:
            st._recent_sweeps[:], st._recent_rounds[:] = [12], [12]
            if st.coords.shape[1] < 80000:
                st.coords = torch.empty((2, 80000, 2), dtype=torch.int32, device="cuda")
        fresh = dev(planes).clone()
        got, _ = st.find(fresh, None, 0.1, 0.9, 0.3, 5, seeds)
        want, _ = ref.find(fresh, None, 0.1, 0.9, 0.3, 5, seeds)
        single, _ = one.find(fresh[:1], None, 0.1, 0.9, 0.3, 5, seeds[:1])
        for p in range(2):
            np.testing.assert_array_equal(got[p][0], want[p][0], err_msg=f"staged call {n} plane {p}")
            np.testing.assert_array_equal(got[p][1], want[p][1], err_msg=f"staged call {n} plane {p}")
        np.testing.assert_array_equal(single[0][0], want[0][0])
    assert st.graph_replays >= 3 and st.graph_captures <= 3, (st.graph_replays, st.graph_captures)
    assert one.graph_replays == 0 and one.graph_captures == 0


def test_edge_grid_chunked_scan(hp):
    """The many-workgroup prefix sum over the cells (d_scan_state) against the one-workgroup one, and the capacity
    guard of the one-call form (phases = 3): a plane with more edges than the list holds reports 0 edges and its
    true count."""
    from magnify_amd import _native as nat

    rng = np.random.default_rng(8)
    P, h, w, grid = 3, 700, 900, 8
    words = 2 * ((h * w + 63) // 64) + 2
    bits = np.zeros((P, words), dtype=np.uint32)
    dens = [0.02, 0.2, 0.0]
    for p in range(P):
        m = (rng.random(h * w) < dens[p])
        packed = np.packbits(m, bitorder="little")
        bits[p].view(np.uint8)[: len(packed)] = packed
    d_bits = dev(bits.view(np.int32))
    n_cells = ((h + grid - 1) // grid) * ((w + grid - 1) // grid)
    out = {}
    for label in ("one", "chunks"):
        counts = torch.zeros((P, n_cells), dtype=torch.int32, device="cuda")
        starts = torch.zeros((P, n_cells), dtype=torch.int32, device="cuda")
        num = torch.zeros((P,), dtype=torch.int32, device="cuda")
        state = torch.zeros((int(nat.lib().mg_edge_grid_scan_words(P, h, w, grid)),), dtype=torch.int64, device="cuda")
        for _ in range(2):  # twice: the second launch meets the first one's published totals
            nat.check(nat.lib().mg_edge_grid(d_bits.data_ptr(), words, P, h, w, grid, counts.data_ptr(), starts.data_ptr(),
                                             num.data_ptr(), 0, 0, state.data_ptr() if label == "chunks" else 0, 0, 1, 0),
                      "mg_edge_grid")
        torch.cuda.synchronize()
        out[label] = (counts.cpu().numpy(), starts.cpu().numpy(), num.cpu().numpy())
    for a, b in zip(out["one"], out["chunks"]):
        np.testing.assert_array_equal(a, b)
    counts, starts, num = out["one"]
    np.testing.assert_array_equal(starts, np.cumsum(counts, axis=1) - counts)
    np.testing.assert_array_equal(num, counts.sum(axis=1))
    assert state.numel() == P * ((n_cells + 4095) // 4096 + 1) and state.numel() > 2 * P  # chunk states + a ticket per plane
    # one call, capacity between the two densities
    cap = int(num[0]) + 10
    coords = torch.full((P, cap, 2), -7, dtype=torch.int32, device="cuda")
    totals = torch.zeros((P,), dtype=torch.int32, device="cuda")
    counts_d, starts_d, num_d = (torch.zeros_like(torch.from_numpy(x)).cuda() for x in out["one"])
    nat.check(nat.lib().mg_edge_grid(d_bits.data_ptr(), words, P, h, w, grid, counts_d.data_ptr(), starts_d.data_ptr(),
                                     num_d.data_ptr(), coords.data_ptr(), cap, state.data_ptr(), totals.data_ptr(), 3, 0),
              "mg_edge_grid")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(totals.cpu().numpy(), num)
    np.testing.assert_array_equal(num_d.cpu().numpy(), [num[0], 0, 0])
    c0 = coords[0, : num[0]].cpu().numpy()
    assert (c0 >= 0).all() and len(np.unique(c0[:, 0].astype(np.int64) * w + c0[:, 1])) == num[0]


def test_stream_probe_moves_the_bytes(hp):
    """mg_stream_probe (the measured streaming ceiling of bench.py): the copy copies, in both launch shapes (a given
    number of workgroups; 0 = one access per lane), the fill writes its pattern, arguments are checked."""
    from magnify_amd import _native as nat

    n = (1 << 20) + 4096 + 16  # not a multiple of a workgroup's 4 KiB
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4 * (n // 4096 + 1), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for blocks in (7, 64, 0):
        dst = torch.zeros_like(src)
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n, 0, sink.data_ptr(), blocks, s), "probe")
        assert torch.equal(src, dst)
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n, 1, sink.data_ptr(), blocks, s), "probe")
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n, 2, sink.data_ptr(), blocks, s), "probe")
        np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint32).reshape(-1, 4), np.tile([1, 2, 3, 4], (n // 16, 1)))
    with pytest.raises(ValueError):
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n - 8, 0, sink.data_ptr(), 8, s), "probe")
    with pytest.raises(ValueError):
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n, 3, sink.data_ptr(), 8, s), "probe")
    with pytest.raises(ValueError):
        nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n, 0, sink.data_ptr(), -1, s), "probe")


def test_flatfield_bound_follows_the_flat_image(hp):
    """The per-chunk bound of the flat image (mg_flatfield_bound) is kept between calls for a caller's float32 device
    tensor and made again when the tensor changes (in place: its version counter; another tensor; a NumPy image): the
    maxima equal the oracle's every time.  Small and large groups (the lanes' shared running maximum)."""
    rng = np.random.default_rng(21)
    tiles = rng.integers(90, 60000, size=(3, 4, 1, 1, 256, 512), dtype=np.uint16)
    d_tiles = dev(tiles)
    flat = vignette((256, 512)).astype(np.float32)
    d_flat = dev(flat)

    def want(flat_img, groups):
        t = np.clip(tiles.astype(np.float64) - 100.0, 0, None).reshape(groups, -1, 256, 512)
        return np.stack([t.max(axis=(1, 2, 3)), (t / flat_img.astype(np.float64)).max(axis=(1, 2, 3))], axis=1)

    for groups in (3, 1, 12):
        got = hp.flatfield_max(d_tiles, d_flat, 100.0, groups).cpu().numpy()
        np.testing.assert_array_equal(got, want(flat, groups))
    calls = []
    orig = hp._call
    hp._call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        hp.flatfield_max(d_tiles, d_flat, 100.0, 3)
        assert calls == ["mg_flatfield_max"]  # the bound of this tensor is in place
        d_flat.mul_(0.5)  # in place: same tensor, new content
        got = hp.flatfield_max(d_tiles, d_flat, 100.0, 3).cpu().numpy()
        assert calls[1:] == ["mg_flatfield_bound", "mg_flatfield_max"]
        np.testing.assert_array_equal(got, want(flat * np.float32(0.5), 3))
        other = dev(np.ascontiguousarray(flat[::-1]))
        got = hp.flatfield_max(d_tiles, other, 100.0, 3).cpu().numpy()
        np.testing.assert_array_equal(got, want(flat[::-1], 3))
        del calls[:]
        got = hp.flatfield_max(d_tiles, flat, 100.0, 3).cpu().numpy()  # a host image: uploaded and bounded afresh
        assert calls == ["mg_flatfield_bound", "mg_flatfield_max"]
        np.testing.assert_array_equal(got, want(flat, 3))
    finally:
        hp._call = orig


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.float32])
def test_plane_minmax_paths(hp, dtype):
    """mg_plane_minmax: the integer fast path (rows of whole 16-byte vectors, four loads in flight, clamped repeats at
    the end), the general path (ragged rows) and planes taken as a strided view of a (T, C, h, w) block."""
    rng = np.random.default_rng(31)
    for shape in ((3, 64, 128), (2, 37, 48), (1, 5, 16), (2, 33, 50), (1, 1, 8)):
        a = (rng.random((shape[0], 2) + shape[1:]) * 250).astype(dtype)
        a[0, 1, shape[1] // 2, shape[2] // 3] = 251  # a unique maximum
        a[-1, 1, 0, shape[2] - 1] = 0
        d = dev(a)
        for ch in (0, 1):
            got = hp.plane_minmax(d[:, ch]).cpu().numpy()
            np.testing.assert_array_equal(got[:, 0], a[:, ch].min(axis=(1, 2)).astype(np.float64))
            np.testing.assert_array_equal(got[:, 1], a[:, ch].max(axis=(1, 2)).astype(np.float64))


@pytest.mark.parametrize("dark", [100.0, 99.5])
def test_flatfield_many_plane_groups(hp, dark):
    """The correction pass with >= 16 plane groups of 8 (the XCD-aware workgroup order: the plane groups of one image part
    run back to back on one XCD) and per-assay maxima, integer-valued and fractional dark (integer-domain / float64
    subtraction), against the oracle: 32 assays x 4 channels, 2 x 2 tiles with overlap."""
    rng = np.random.default_rng(17)
    n_t, n_c, ty, tx = 32, 4, 40, 1032
    tiles = rng.integers(60, 60000, size=(n_t, n_c, 2, 2, ty, tx), dtype=np.uint16)
    tiles[3] = 100  # an assay at the dark level: maxima 0
    flat = vignette((ty, tx))
    for overlap in (0, 8, 16):  # aligned kernel (remapped grid), general kernel, aligned with cropping
        img, minmax = hp.flatfield_stitch(dev(tiles), overlap, flat, dark, n_groups=n_t)
        got = img.cpu().numpy()
        for t in range(n_t):
            want = rp.stitch(rp.flatfield_correct(tiles[t][None], flat, dark), overlap)[0]
            np.testing.assert_array_equal(got[t], want, err_msg=f"assay {t} overlap {overlap}")
            mm = minmax.cpu().numpy().reshape(n_t, n_c, 2)[t]
            np.testing.assert_array_equal(mm[:, 0], want.min(axis=(-1, -2)))
            np.testing.assert_array_equal(mm[:, 1], want.max(axis=(-1, -2)))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float64])
def test_masked_median_types_and_time_masks(hp, dtype):
    """mg_roi_masked_median (identify.py:76-80, filter.py:20-22, 74, 82): every roi dtype the path carries, a mask per
    timepoint (a chip searched at several timesteps), one mask for all, a broadcast view -- against numpy's nanmedian on
    the oracle's float64 copy (rp.roi_reduce), exactly.  Floats: negative values, +-0, infinities, NaN pixels (ignored,
    as nanmedian ignores them), near-ties that differ in the last bit; even and odd counts; empty masks -> NaN."""
    rng = np.random.default_rng(11)
    m, c, t, L = 9, 3, 4, 17
    if np.dtype(dtype).kind == "f":
        roi = rng.normal(0, 50, size=(m, c, t, L, L)).astype(dtype)
        roi[0] = np.where(rng.random((c, t, L, L)) < 0.5, dtype(7.25), np.nextafter(dtype(7.25), dtype(8)))  # one ulp apart
        roi[1, :, :, ::3] = np.nan
        roi[2, 0, 0, :4, :4] = [[-0.0, 0.0, np.inf, -np.inf]] * 4
        roi[3] = np.abs(roi[3]) * 1e30 if dtype == np.float32 else np.abs(roi[3]) * 1e300
        roi[4] = -np.abs(roi[4])
    else:
        hi = 255 if dtype == np.uint8 else 65535
        roi = rng.integers(0, hi + 1, size=(m, c, t, L, L)).astype(dtype)
        roi[0] = rng.integers(100, 103, size=(c, t, L, L))  # many ties
        roi[3] = hi
        roi[4] = 0
    fg = rng.random((m, t, L, L)) < 0.4
    bg = ~fg & (rng.random((m, t, L, L)) < 0.5)
    fg[5] = False                       # empty masks
    bg[5, 1] = False
    fg[6, :, :, :] = False
    fg[6, :, 3, 3] = True               # a single pixel
    fg[7, 2] = False
    fg[7, 2, :2, :1] = True             # two pixels: the mean of both
    red = rp.roi_reduce(roi, fg, bg)
    d_roi = dev(roi)
    for name, mask in (("fg", fg), ("bg", bg)):
        got = hp.masked_median(d_roi, dev(mask.view(np.uint8))).cpu().numpy()
        np.testing.assert_array_equal(got, red[f"{name}_median"], err_msg=f"{name} per-time masks")
    # one mask for all timepoints: as a 3-D mask, as (m, 1, L, L) and as an expanded (stride-0) view
    one = fg[:, :1]
    want = rp.roi_reduce(roi, np.broadcast_to(one, fg.shape), bg, medians=True)["fg_median"]
    d_one = dev(one.view(np.uint8))
    for mask in (d_one[:, 0], d_one, d_one.expand(m, t, L, L), d_one.bool()):
        np.testing.assert_array_equal(hp.masked_median(d_roi, mask).cpu().numpy(), want)
    if dtype == np.uint16:  # the round-1 entry point is the same kernel
        np.testing.assert_array_equal(hp.masked_median_u16(d_roi, d_one[:, 0].contiguous()).cpu().numpy(), want)
    with pytest.raises(ValueError):
        hp.masked_median(d_roi, dev(fg[:, :2].view(np.uint8)))
    assert hp.masked_median(d_roi[:0], dev(fg[:0].view(np.uint8))).shape == (0, c, t)


def test_graph_survives_regrown_buffers(hp):
    """a captured chain bakes in the addresses AND the capacity of the output set, the claim ring, the
    window histogram.  When one of them is made anew between two calls (more circles than the set holds; another
    suppression distance) no old graph may be replayed -- whatever address the allocator hands the new buffer.
    Compared call by call with a finder that launches eagerly."""
    shape = (320, 384)
    images = [np.stack([noisy_bead_image(900 + 10 * n + j, shape, 6)[0] for j in range(2)]) for n in range(12)]
    buf = torch.empty((2,) + shape, dtype=torch.uint16, device="cuda")
    cf = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    ref = hp.CircleFinder(2, shape[0], shape[1], 5, 21, 60000)
    ref._graphs = None
    min_dist = 5
    for n, planes in enumerate(images):
        buf.copy_(dev(planes))
        if n > 0:
            cf._recent_sweeps[:], cf._recent_rounds[:] = [12], [12]
            if cf.coords.shape[1] < 80000:
                cf.coords = torch.empty((2, 80000, 2), dtype=torch.int32, device="cuda")
        if n == 5:  # the output sets are dropped and made anew, larger (what a call with more circles does) ...
            captured = cf.graph_captures
            assert cf._graphs, "nothing was captured before the regrow"
            old_cap = cf._out_cap
            cf._out_buffers(3 * cf._out_cap)
            assert cf._out_cap > old_cap and not cf._graphs  # ... and every graph went with them
            torch.cuda.empty_cache()  # the freed sets' addresses are up for grabs
        if n == 9:
            min_dist = 7  # another claim ring: uploaded anew
        seeds = [8000 + 2 * n, 8001 + 2 * n]
        got, _ = cf.find(buf, None, 0.1, 0.9, 0.3, min_dist, seeds, stable_input=True)
        want, _ = ref.find(buf, None, 0.1, 0.9, 0.3, min_dist, seeds)
        for p in range(2):
            np.testing.assert_array_equal(got[p][0], want[p][0], err_msg=f"call {n} plane {p}")
            np.testing.assert_array_equal(got[p][1], want[p][1], err_msg=f"call {n} plane {p}")
    assert "graph_error" not in cf.stats, cf.stats.get("graph_error")
    assert cf.graph_captures > captured and cf.graph_replays >= 4, (cf.graph_captures, cf.graph_replays, cf.calls)


@pytest.mark.parametrize("num_iter", [3000, 200000])
def test_candidates_degenerate_triplets(hp, num_iter):
    """mg_candidate_keys on edge maps made of exactly the triplets that stress its arithmetic (VERDICT r3 item 3a;
    utils.py:319-342): lone pixels (p1 = p2 = p0: 0 / eps), horizontal runs (d_row = 0: slopes of 1e20), vertical runs
    (both slopes 0: the divisor is eps itself), diagonals (collinear: equal slopes, centres at 1e20 and beyond float32),
    two-pixel cells, and a dense block where general triplets mix with all of these.  The round-4 kernel reads slope
    and intercept from a per-workgroup LDS table and divides without range scaling; the raw float32 triples must equal
    the oracle's BIT FOR BIT, NaN and infinities included, with fewer edges than iterations (strata of one edge: the
    first hash is never drawn) and with more (num_iter 3000 < edges), and the keys must be those of the triples."""
    from magnify_amd import _native as nat

    h, w, grid, min_r, max_r = 200, 260, 20, 5, 14
    e = np.zeros((3, h, w), dtype=np.uint8)
    e[0, 10::40, 10::40] = 1                      # lone pixels
    e[0, 5, 60:80] = 1                            # a horizontal run inside one cell row
    e[0, 40:60, 7] = 1                            # a vertical run
    e[0, np.arange(100, 120), np.arange(100, 120)] = 1   # a diagonal: collinear triplets
    e[0, np.arange(140, 160), np.arange(59, 39, -1)] = 1  # the other diagonal
    e[0, 181, 181] = e[0, 183, 190] = 1           # two pixels in a cell
    rng = np.random.default_rng(5)
    e[1] = rng.random((h, w)) < 0.2               # dense: general triplets (and every coincidence among them)
    yy, xx = np.mgrid[0:h, 0:w]
    for cy, cx, r in ((60, 70, 9), (120, 180, 12), (150, 60, 6)):
        e[2] |= (np.abs(np.hypot(yy - cy, xx - cx) - r) < 0.6).astype(np.uint8)  # true circles
    e[2, 0, :] = e[2, :, 0] = e[2, h - 1, :] = e[2, :, w - 1] = 1  # the frame: centres off the image
    P = e.shape[0]
    gr, gc = -(-h // grid), -(-w // grid)
    lists = [rn.grid_array(e[k], grid) for k in range(P)]
    cap = max(len(c) for c, _, _ in lists)
    coords = np.zeros((P, cap, 2), dtype=np.int32)
    starts = np.zeros((P, gr, gc), dtype=np.int32)
    counts = np.zeros((P, gr, gc), dtype=np.int32)
    for k, (c, s, n) in enumerate(lists):
        coords[k, : len(c)], starts[k], counts[k] = c, s, n
    n_edges = np.array([len(c) for c, _, _ in lists], dtype=np.int32)
    seeds = np.array([101, 102, 103], dtype=np.uint64)
    d = {k: dev(v) for k, v in dict(coords=coords, starts=starts, counts=counts, n_edges=n_edges).items()}
    d_seeds = torch.from_numpy(seeds.view(np.int64)).cuda()
    keys = torch.empty((P, num_iter), dtype=torch.int32, device="cuda")
    raw = torch.empty((P, num_iter, 3), dtype=torch.float32, device="cuda")
    nat.check(nat.lib().mg_candidate_keys(d["coords"].data_ptr(), cap, d["starts"].data_ptr(), d["counts"].data_ptr(),
                                          d["n_edges"].data_ptr(), P, h, w, grid, d_seeds.data_ptr(), num_iter, min_r, max_r,
                                          keys.data_ptr(), raw.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "mg_candidate_keys")
    got, got_keys = raw.cpu().numpy(), keys.cpu().numpy().view(np.uint32)
    ntc = (w + 2 * max_r + 63) // 64
    kinds = set()
    for k in range(P):
        picks = rn.draw_picks(int(seeds[k]), num_iter, e[k], grid)
        want = rn.candidate_circles_from_picks(e[k], grid, *picks)
        np.testing.assert_array_equal(got[k].view(np.uint32), want.view(np.uint32), err_msg=f"plane {k}")
        kinds |= {"nan"} if np.isnan(want).any() else set()
        kinds |= {"inf"} if np.isinf(want).any() else set()
        kinds |= {"zero radius"} if (want[:, 2] == 0).any() else set()
        kinds |= {"huge"} if (np.abs(want[np.isfinite(want)]) > 1e15).any() else set()  # centres of collinear triplets
        # the keys: the reference's radius / on-image filter (utils.py:157-166) on the rounded triples
        with np.errstate(invalid="ignore"):
            ok = (want[:, 2] >= min_r) & (want[:, 2] <= max_r) & (np.abs(np.round(want[:, 0])) < 1e9) & (np.abs(np.round(want[:, 1])) < 1e9)
            c = np.where(ok[:, None], np.round(want), 0).astype(np.int64)
        ok &= (c[:, 0] + c[:, 2] >= 0) & (c[:, 1] + c[:, 2] >= 0) & (c[:, 0] - c[:, 2] < h) & (c[:, 1] - c[:, 2] < w)
        pr, pc = c[:, 0] + max_r, c[:, 1] + max_r
        key = (((pr >> 6) * ntc + (pc >> 6)) << 17) | ((c[:, 2] - min_r) << 12) | ((pr & 63) << 6) | (pc & 63)
        np.testing.assert_array_equal(got_keys[k], np.where(ok, key, 0xFFFFFFFF).astype(np.uint32), err_msg=f"keys of plane {k}")
        if k == 2:
            assert ok.sum() > num_iter // 50  # the circles are found
    assert {"zero radius", "huge"} <= kinds and ("nan" in kinds or "inf" in kinds), kinds


@pytest.mark.parametrize("L,time_major", [(100, False), (40, False), (64, True)])
def test_roi_image_centric_pass_equals_window_pass(hp, monkeypatch, L, time_major):
    """The image-centric ROI pass of round 4 (MG_ROI_TILES=1: a workgroup owns a 16 x 384 tile, loads its planes into
    LDS once and serves every window's fragment -- roi pixels, mask bytes, sums and counts by atomic adds) against the
    window-centric pass and against the masks of the oracle's circle_labels map: windows cut by tile borders in both
    directions, shifted into the image at its edges, overlapping and contested disks, a crowded assay (hundreds of
    windows per tile: several rounds), an empty one, 5 planes per assay (the second pass over the planes is partial)."""
    rng = np.random.default_rng(44)
    c, t, h, w = (5, 1, 150, 800) if not time_major else (2, 3, 150, 800)
    A = 3
    images = rng.integers(0, 65536, size=(A, t, c, h, w) if time_major else (A, c, t, h, w)).astype(np.uint16)
    crowded = np.column_stack([rng.integers(0, h, 300), rng.integers(300, 500, 300), rng.integers(2, 14, 300)])
    assays = [np.array([[20, 20, 8], [5, 790, 6], [149, 400, 10], [80, 380, 9], [84, 392, 9], [16, 384, 5], [15, 383, 12],
                        [31, 767, 7], [-3, 100, 6], [90, 805, 7], [80, 84, 25], [75, 300, 1]]), crowded, np.empty((0, 3), dtype=np.int64)]
    tab = np.zeros((A, 320, 3), dtype=np.int32)
    for a, beads in enumerate(assays):
        tab[a, : len(beads)] = beads
    counts = [len(b) for b in assays]
    kw = dict(disks=True, device_tables=(dev(tab), counts, 25), time_major=time_major)
    monkeypatch.delenv("MG_ROI_TILES", raising=False)
    want = hp.roi_gather_reduce(dev(images), None, L, None, **kw)
    monkeypatch.setenv("MG_ROI_TILES", "1")
    got = hp.roi_gather_reduce(dev(images), None, L, None, **kw)
    monkeypatch.delenv("MG_ROI_TILES")
    for key in ("roi", "fg", "bg", "sums", "counts"):
        np.testing.assert_array_equal(got[key].cpu().numpy(), want[key].cpu().numpy(), err_msg=key)
    off = got["offsets"]
    lab = rn.circle_labels(assays[0][assays[0][:, 2] >= 2], h, w)  # (a radius below 2 covers nothing: undefined in the reference)
    for i, (row, col, r) in enumerate(assays[0]):
        top, bottom, left, right = rn.bounding_box(int(col), int(row), L, w, h)
        sub = lab[top:bottom, left:right]
        if r >= 2:
            np.testing.assert_array_equal(got["fg"][off[0] + i].cpu().numpy().astype(bool), sub == i)
        np.testing.assert_array_equal(got["bg"][off[0] + i].cpu().numpy().astype(bool), sub == -1)
        win = images[0][..., top:bottom, left:right]
        win = win.transpose(1, 0, 2, 3) if time_major else win
        np.testing.assert_array_equal(got["roi"][off[0] + i].cpu().numpy(), win)
    # only the reductions (no pixel stack, no mask bytes)
    monkeypatch.setenv("MG_ROI_TILES", "1")
    light = hp.roi_gather_reduce(dev(images), None, L, None, want_roi=False, want_masks=False, **kw)
    monkeypatch.delenv("MG_ROI_TILES")
    np.testing.assert_array_equal(light["sums"].cpu().numpy(), want["sums"].cpu().numpy())
    np.testing.assert_array_equal(light["counts"].cpu().numpy(), want["counts"].cpu().numpy())


@pytest.mark.parametrize("device_counts", [False, True])
def test_roi_windows_visited_band_by_band(hp, monkeypatch, device_counts):
    """mg_roi_window_order: every assay's markers band by band (64 rows) and left to right, equal keys in table order, the
    markers beyond the launch's bound left out; and the ROI pass that visits its windows in that order writes what the
    pass in table order writes (compact and padded bead tables, an empty assay, offsets made on the device)."""
    from magnify_amd import _native as nat

    rng = np.random.default_rng(45)
    c, t, h, w, L, A = 3, 2, 400, 640, 50, 4
    images = rng.integers(0, 65536, size=(A, c, t, h, w)).astype(np.uint16)
    assays = [np.column_stack([rng.integers(-5, h + 5, n), rng.integers(-5, w + 5, n), rng.integers(2, 12, n)]) for n in (700, 0, 33, 1)]
    assays[0][100:140] = assays[0][100]  # equal keys
    cap = 720
    tab = np.zeros((A, cap, 3), dtype=np.int32)
    for a, beads in enumerate(assays):
        tab[a, : len(beads)] = beads
    counts = [len(b) for b in assays]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    m = int(off[-1])
    lib, s = nat.lib(), torch.cuda.current_stream().cuda_stream
    for bound in (m, m - 10, m + 50):
        order = torch.full((bound,), -1, dtype=torch.int32, device="cuda")
        nat.check(lib.mg_roi_window_order(dev(tab).data_ptr(), cap, dev(off).data_ptr(), A, bound, order.data_ptr(), s), "order")
        got = order.cpu().numpy()
        for a, beads in enumerate(assays):
            lo, hi = int(off[a]), min(int(off[a + 1]), bound)
            if hi <= lo:
                continue
            b = beads[: hi - lo]
            key = (np.clip(b[:, 0], 0, None) >> 6) * (1 << 17) + np.clip(b[:, 1], 0, None)
            np.testing.assert_array_equal(got[lo:hi], lo + np.argsort(key, kind="stable"))
        assert (got[m:] == -1).all()
    kw = dict(disks=True, device_tables=(dev(tab), counts if not device_counts else None, 11))
    if device_counts:
        kw["device_counts"] = (dev(np.asarray(counts, dtype=np.int32)), cap, None)
    monkeypatch.setattr(hp, "_ROI_ORDER_MIN", 1)
    monkeypatch.setattr(hp, "_ROI_ORDER", False)
    want = hp.roi_gather_reduce(dev(images), None, L, None, **kw)
    monkeypatch.setattr(hp, "_ROI_ORDER", True)
    got = hp.roi_gather_reduce(dev(images), None, L, None, **kw)
    for key in ("roi", "fg", "bg", "sums", "counts"):
        np.testing.assert_array_equal(got[key].cpu().numpy()[:m], want[key].cpu().numpy()[:m], err_msg=key)
    # the compact table of the host-side route
    want = hp.roi_gather_reduce(dev(images), assays, L, None, disks=True)
    monkeypatch.setattr(hp, "_ROI_ORDER", False)
    plain = hp.roi_gather_reduce(dev(images), assays, L, None, disks=True)
    for key in ("roi", "fg", "bg", "sums", "counts"):
        np.testing.assert_array_equal(plain[key].cpu().numpy(), want[key].cpu().numpy(), err_msg=key)
