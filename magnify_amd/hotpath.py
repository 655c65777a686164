"""Device-side orchestration of the hot path on PyTorch-ROCm tensors.

PyTorch is plumbing only (device memory, streams); every numeric step is a hand-written HIP
kernel reached through the C ABI (``include/magnify_hip.h``).  Host code here mirrors the control
flow of the reference's ``utils.find_circles`` (src/magnify/utils.py:102-222),
``flatfield_correct`` (preprocess.py:62-88), ``Stitcher`` (stitch.py:12-46) and the ROI part of
``BeadFinder.__call__`` (find.py:561-602), batched over planes.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _native as nat

GRID_LENGTH = 20  # find.py:215, 348, 482


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    """Raw handle of torch's current stream on the current device (the short way: ``torch.cuda.current_stream()``
    builds a Stream object, several microseconds per C-ABI call)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class StageTimer:
    """Per-stage HIP-event timing on the stream the kernels are launched on (torch's current
    stream): ``with timer.stage("name"):`` around C-ABI calls; ``timer.summary()`` after a sync."""

    def __init__(self, allow_graphs=False):
        """``allow_graphs``: the finder may replay its optimistic chain as one hipGraph launch -- the stages inside it
        are then not timed (no events inside a graph), everything launched around it (flat-field, ROI pass) still is."""
        self.events = []
        self.allow_graphs = allow_graphs

    class _Ctx:
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()

        def __exit__(self, *exc):
            self.end.record()
            self.timer.events.append((self.name, self.start, self.end))

    def stage(self, name):
        return StageTimer._Ctx(self, name)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.events:
            tot, cnt = out.get(name, (0.0, 0))
            out[name] = (tot + a.elapsed_time(b), cnt + 1)
        return out


class _NoTimer:
    class _Ctx:
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    _ctx = _Ctx()

    def stage(self, name):
        return self._ctx


TIMER = _NoTimer()


def set_timer(timer):
    """Install a StageTimer (or None) used by every hot-path call."""
    global TIMER
    TIMER = timer if timer is not None else _NO_TIMER


_NO_TIMER = TIMER


_SYNC_EVERY_CALL = bool(os.environ.get("MG_SYNC_CALLS"))
_CAPTURING = False  # inside a stream capture (CircleFinder._capture_chain): no timing events


def _call(name, *args, stage=None):
    """One C-ABI call: timed (if a StageTimer is installed) and status-checked."""
    if TIMER is _NO_TIMER or _CAPTURING:
        rc = getattr(nat.lib(), name)(*args)
        if rc:
            nat.check(rc, name)
        if _SYNC_EVERY_CALL and not _CAPTURING:  # MG_SYNC_CALLS=1: a kernel that faults is named by the call it came from
            import sys

            print(f"[mg] {name}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
        return
    with TIMER.stage(stage or name):
        nat.check(getattr(nat.lib(), name)(*args), name)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("magnify_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
    nat.lib()


# --------------------------------------------------------------------------------------
# A2 + A1: flat-field correction fused with stitching
# --------------------------------------------------------------------------------------


def _df_operand(value, ty, tx, device):
    """Scalar -> (float, None, 0); image -> (0.0, device tensor, dtype code)."""
    if np.isscalar(value) or (isinstance(value, np.ndarray) and value.ndim == 0):
        return float(value), None, 0
    arr = np.asarray(value) if not isinstance(value, torch.Tensor) else value
    if isinstance(arr, np.ndarray):
        if arr.shape != (ty, tx):
            raise ValueError(f"flat/dark field image must have the tile shape {(ty, tx)}, got {arr.shape}")
        arr = arr.astype(np.float32 if arr.dtype == np.float32 else np.float64)
        arr = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    else:
        if tuple(arr.shape) != (ty, tx):
            raise ValueError(f"flat/dark field image must have the tile shape {(ty, tx)}")
        arr = arr.to(device=device, dtype=torch.float32 if arr.dtype == torch.float32 else torch.float64).contiguous()
    return 0.0, arr, nat.dtype_code(arr.dtype)


def flatfield_is_identity(dtype, flatfield=1.0, darkfield=0.0) -> bool:
    """Integer pixels with the reference's default operands (flat 1, dark 0): the correction leaves every pixel as
    it is (mg_flatfield_is_identity), so neither pass 1 nor the arithmetic of pass 2 is needed."""
    scalar = lambda v: np.isscalar(v) or (isinstance(v, np.ndarray) and v.ndim == 0)  # noqa: E731
    if not (scalar(flatfield) and scalar(darkfield)):
        return False
    return bool(nat.lib().mg_flatfield_is_identity(nat.dtype_code(dtype), float(darkfield), 0, float(flatfield), 0))


def flatfield_max(tiles: torch.Tensor, flatfield=1.0, darkfield=0.0, n_groups=1) -> torch.Tensor:
    """Pass 1 of flatfield_correct: the two global maxima as a float64 (n_groups, 2) device tensor
    (``n_groups`` equal blocks along the leading axes: one per independent assay)."""
    require_gpu()
    ty, tx = tiles.shape[-2:]
    tiles = tiles.contiguous()
    dk, dkt, dkc = _df_operand(darkfield, ty, tx, tiles.device)
    fl, flt, flc = _df_operand(flatfield, ty, tx, tiles.device)
    max2 = torch.full((n_groups, 2), -math.inf, dtype=torch.float64, device=tiles.device)
    n_tiles = tiles.numel() // (ty * tx)
    code = nat.dtype_code(tiles.dtype)
    f32_image = flt is not None and flt.dtype == torch.float32 and dkt is None and (ty * tx) % (16 // tiles.element_size()) == 0
    words = int(nat.lib().mg_flatfield_max_scratch_floats(code, ty, tx)) if f32_image else 0
    scratch = None
    if words > 0:
        # the per-chunk bound of the flat image (mg_flatfield_bound), computed when the image changes, not per call:
        # a caller's device tensor is recognised by identity + version counter (the entry holds a reference to it);
        # anything converted or uploaded on the way in (NumPy, float64, host tensors) is bounded afresh.
        # (one block per stream: the sub-batch threads of a multi-stream StackProcessor fill it side by side)
        name = f"flatfield_max_scratch@{_stream()}"
        scratch = pooled(name, words, (), torch.float32, tiles.device)
        same = flt is flatfield
        key = (id(flatfield), flatfield._version, flt.data_ptr(), scratch.data_ptr(), code, ty, tx) if same else None
        if key is None or _BOUND_OF.get(name, (None, None))[0] != key:
            _call("mg_flatfield_bound", flt.data_ptr(), flc, code, ty, tx, scratch.data_ptr(), words, _stream())
            _BOUND_OF[name] = (key, flatfield if same else None)
    _call("mg_flatfield_max", tiles.data_ptr(), code, n_tiles, n_groups, ty, tx, dk, _ptr(dkt),
                                         dkc, fl, _ptr(flt), flc, max2.data_ptr(), _ptr(scratch), words, _stream())
    return max2


_BOUND_OF = {}  # scratch block -> (key of the flat image it bounds, the image)


def flatfield_stitch(tiles: torch.Tensor, overlap: int, flatfield=1.0, darkfield=0.0, apply_flatfield=True,
                     max2: torch.Tensor | None = None, want_minmax=True, out: torch.Tensor | None = None,
                     minmax_out: torch.Tensor | None = None, n_groups=1):
    """tiles (C, T, R, Cc, ty, tx) -> image (C, T, R*hy, Cc*hx) and per-plane min/max (C*T, 2).

    ``max2`` lets a multi-GPU caller supply all-reduced maxima."""
    require_gpu()
    if overlap < 0:
        raise ValueError("Overlap must be non-negative.")
    c, t, nr, nc, ty, tx = tiles.shape
    if overlap >= ty or overlap >= tx:
        raise ValueError(f"Overlap ({overlap}) must be smaller than tile size ({ty}x{tx}).")
    tiles = tiles.contiguous()
    clip, rem = overlap // 2, overlap % 2
    hy, hx = ty - 2 * clip - rem, tx - 2 * clip - rem
    dk, dkt, dkc = _df_operand(darkfield, ty, tx, tiles.device)
    fl, flt, flc = _df_operand(flatfield, ty, tx, tiles.device)
    if (c * t) % n_groups:
        raise ValueError("n_groups must divide the number of planes")
    if apply_flatfield and flatfield_is_identity(tiles.dtype, flatfield, darkfield):
        apply_flatfield = False
    if apply_flatfield and max2 is None:
        max2 = flatfield_max(tiles, flatfield, darkfield, n_groups)
    if out is None:
        image = torch.empty((c, t, nr * hy, nc * hx), dtype=tiles.dtype, device=tiles.device)
    else:
        image = out
        assert image.is_contiguous() and image.numel() == c * t * nr * hy * nc * hx and image.dtype == tiles.dtype
    minmax = None
    if want_minmax:
        if minmax_out is None:
            minmax = torch.empty((c * t, 2), dtype=torch.float64, device=tiles.device)
        else:
            minmax = minmax_out
            assert minmax.is_contiguous() and minmax.numel() == c * t * 2 and minmax.dtype == torch.float64
        minmax.view(-1, 2).copy_(_minmax_init(c * t, tiles.device))  # (+inf, -inf) rows: one copy, not two fills
    _call("mg_flatfield_apply_stitch", tiles.data_ptr(), nat.dtype_code(tiles.dtype), c * t, nr, nc, ty, tx,
                                                  overlap, int(bool(apply_flatfield)), (c * t) // n_groups, dk, _ptr(dkt), dkc, fl,
                                                  _ptr(flt), flc, _ptr(max2), image.data_ptr(), _ptr(minmax),
                                                  _stream())
    return image, minmax


_MINMAX_INIT = {}


def _minmax_init(n, device):
    """(n, 2) float64 rows (+inf, -inf) on the device: what the min / max atomics start from."""
    key = (int(n), str(device))
    t = _MINMAX_INIT.get(key)
    if t is None:
        if len(_MINMAX_INIT) > 16:
            _MINMAX_INIT.clear()
        t = torch.tensor([math.inf, -math.inf], dtype=torch.float64, device=device).repeat(int(n), 1)
        _MINMAX_INIT[key] = t
    return t


def plane_minmax(planes: torch.Tensor) -> torch.Tensor:
    """Per-plane (min, max) of a (P, H, W) view with uniform plane stride."""
    require_gpu()
    p, h, w = planes.shape
    assert planes.stride(2) == 1 or w <= 1
    out = _minmax_init(p, planes.device).clone()
    _call("mg_plane_minmax", planes.data_ptr(), nat.dtype_code(planes.dtype), p, planes.stride(0), h, w,
                                        planes.stride(1), out.data_ptr(), _stream())
    return out


# --------------------------------------------------------------------------------------
# np.quantile on the float32 gradient magnitude, from the integer histogram
# --------------------------------------------------------------------------------------


def _grad_of(m: int) -> np.float32:
    # utils.py:120: sqrt(dx**2 + dy**2) in float32; the float32 sum of two exact squares is the
    # correctly rounded float32 of the integer m
    return np.sqrt(np.float32(m))


def quantile_indexes(n: int, q: float):
    """Indexes and weight that numpy (2.x) uses for np.quantile(float32_array, q),
    method 'linear': a Python-float q is cast to the array dtype, so the virtual index is
    float32 arithmetic (numpy/lib/_function_base_impl.py: quantile, _QuantileMethods["linear"],
    _get_indexes, _get_gamma)."""
    qf = np.float32(q)
    vi = (n - 1) * qf  # float32: lambda n, quantiles: (n - 1) * quantiles
    prev = np.floor(vi)
    nxt = prev + 1
    if vi >= n - 1:
        prev_i = nxt_i = n - 1
    elif vi < 0:
        prev_i = nxt_i = 0
    else:
        prev_i, nxt_i = int(prev), int(nxt)
    gamma = np.float32(vi - prev)
    return prev_i, nxt_i, gamma


def lerp_f32(a: np.float32, b: np.float32, t: np.float32) -> np.float32:
    """numpy's _lerp in float32."""
    diff = np.float32(b - a)
    if t >= 0.5:
        return np.float32(b - diff * np.float32(1 - t))
    return np.float32(a + diff * t)


def canny_int_thresholds(lo: float, hi: float):
    """Threshold preparation of cv::Canny(dx, dy, t1, t2, L2gradient=true) (call site utils.py:128-134)."""
    lo, hi = float(lo), float(hi)
    if lo > hi:
        lo, hi = hi, lo
    lo, hi = min(32767.0, lo), min(32767.0, hi)
    if lo > 0:
        lo *= lo
    if hi > 0:
        hi *= hi
    return int(math.floor(lo)), int(math.floor(hi))


COARSE_SHIFT = 13
COARSE_BINS = 4096  # covers m < 2**25 (max 2 * 4080**2 = 33 292 800)
FINE_BINS = 1 << COARSE_SHIFT
COMBINED_BINS = FINE_BINS + COARSE_BINS


# --------------------------------------------------------------------------------------
# find_circles, batched over planes
# --------------------------------------------------------------------------------------


class CircleFinder:
    """``utils.find_circles`` (utils.py:102-222) for a batch of equally sized planes.

    Workspace tensors are allocated once per (P, H, W, radii, num_iter) and reused.
    After ``find`` the intermediate device tensors (``blur``, ``edges``, ``angle``,
    ``coords`` ...) stay available for inspection by the parity tests.
    """

    MAX_GROUP = 64  # most hysteresis sweeps / suppression rounds launched between two host checks
    # hipGraph captures per finder: 12 to start with, one more per 64 optimistic calls after that (a capture costs ~9 ms:
    # a caller whose launch sequences never repeat pays at most ~0.15 ms a call for trying; one whose hints drift over a
    # long run keeps getting graphs)
    MAX_CAPTURES = 12

    def __init__(self, n_planes, h, w, min_radius, max_radius, num_iter, device="cuda", grid_length=GRID_LENGTH):
        require_gpu()
        if min_radius > max_radius:
            raise ValueError("min_radius must be <= max_radius")
        self.P, self.h, self.w = int(n_planes), int(h), int(w)
        self.min_r, self.max_r, self.num_iter = int(min_radius), int(max_radius), int(num_iter)
        self.grid = int(grid_length)
        self.dev = torch.device(device)
        P, dev = self.P, self.dev
        self.gr, self.gc = math.ceil(h / self.grid), math.ceil(w / self.grid)
        self.n_cells = self.gr * self.gc
        i32, u8 = torch.int32, torch.uint8
        self.blur = torch.empty((P, h, w), dtype=u8, device=dev)
        self.edges = None  # {0,1} byte map, only with keep_debug_maps
        self.angle = None  # float32 (P, h, w), valid at edge pixels: only when something reads it (see need_angle_map)
        self.hist = torch.zeros((P, COMBINED_BINS), dtype=i32, device=dev)
        self.hist_base = torch.zeros((P,), dtype=i32, device=dev)
        # per-workgroup histogram slots: plain stores + a reduce kernel instead of global atomics
        self.hist_scratch = torch.empty((int(nat.lib().mg_blur_hist_scratch_words(P, h, w)),), dtype=i32, device=dev)
        self.thresh = torch.zeros((P, 2), dtype=i32, device=dev)
        self.quant_d = torch.zeros((P, 2), dtype=torch.float32, device=dev)  # np.quantile's two values per plane
        # Every per-plane counter the host ever looks at lives in ONE block: one fill clears it, one copy fetches it
        # (rows: the eight counters, then `changed` per hysteresis sweep, then `undecided` per suppression round).
        G = self.MAX_GROUP
        self.status = torch.zeros((8 + 2 * G, P), dtype=i32, device=dev)
        (self.num_edges, self.edge_totals, self.unresolved, self.num_alive, self.num_out, self.num_scored,
         self.num_surv, self.num_circles) = (self.status[k] for k in range(8))
        self.changed, self.undecided = self.status[8: 8 + G], self.status[8 + G: 8 + 2 * G]
        self.status_host = torch.zeros((8 + 2 * G, P), dtype=i32).pin_memory()
        self._quantiles = None
        # window passes of the threshold search (ranks inside coarse histogram bins), all on the device: per-plane state
        # block, the base of the next pass, the window histogram (allocated when a shape first needs a pass)
        self.win_state = torch.zeros((P, 16), dtype=i32, device=dev)
        self.hist_win = None
        self._recent_win = []
        tx, ty = nat.C.c_int(0), nat.C.c_int(0)
        nat.check(nat.lib().mg_hysteresis_tiles(h, w, nat.C.byref(tx), nat.C.byref(ty)), "mg_hysteresis_tiles")
        # active-tile flags of the hysteresis sweeps, one layer per sweep of a group (+ the last layer of the group before)
        self.tile_flags = torch.zeros((G + 1, P, ty.value, tx.value), dtype=u8, device=dev)
        self.scan_state = torch.zeros((max(1, int(nat.lib().mg_edge_grid_scan_words(P, h, w, self.grid))),),
                                      dtype=torch.int64, device=dev)
        # Optimistic chain (see find): sweeps / rounds / capacities taken from the calls before, everything checked
        # at the one host round trip at the end.  MG_CHECKED_CHAIN=1 keeps the three-round-trip chain.
        self.optimistic = not os.environ.get("MG_CHECKED_CHAIN")
        self._recent_sweeps, self._recent_rounds = [], []
        # how the calls of this finder went (find); followed_again: optimistic calls whose follow-up pass ran twice
        self.calls = {"optimistic": 0, "repaired": 0, "checked": 0, "followed_again": 0}
        self._out_sets, self._out_turn, self._out_cap = [None, None], 0, 0
        self._n_collects, self.follow_result = 0, None
        # hipGraphs of the optimistic chain (_optimistic_chain), by launch-sequence key
        self._graphs = None if os.environ.get("MG_NO_GRAPH") else {}
        self.graph_replays, self.graph_captures = 0, 0
        self._graph_seen, self._in_stage, self._retired_graphs = set(), None, []
        self._round_spare = 0
        self._mm = torch.empty((P, 2), dtype=torch.float64, device=dev)
        self.words = 2 * ((h * w + 63) // 64) + 2  # bitmap words per plane (even, one spare)
        self.edge_bits = torch.zeros((P, self.words), dtype=i32, device=dev)  # strong bits = edges
        self.weak_bits = torch.zeros((P, self.words), dtype=i32, device=dev)
        # gradient orientation bins (three bit planes per image plane) for the scoring prefilter
        self.class_bits = torch.zeros((P, 3, self.words), dtype=i32, device=dev)
        self.keep_debug_maps = False  # True: also produce the {0,1} byte map and the angle map (tests)
        self.cell_counts = torch.zeros((P, self.n_cells), dtype=i32, device=dev)
        self.cell_starts = torch.zeros((P, self.n_cells), dtype=i32, device=dev)
        self.coords = None
        ntr, ntc, self.n_layers, self.bitmap_words = nat.dedup_layout(h, w, self.min_r, self.max_r)
        self.n_tiles = ntr * ntc
        self.bitmap = self.layer_offsets = None  # only the atomicOr path needs them (allocated on first use)
        # keyed de-duplication (mg_candidate_keys + mg_keys_to_circles: no global atomics) whenever its
        # 32-bit key has room for the layout; else the atomicOr bitmap path
        self.keyed = (ntr * ntc < 32768 and self.max_r - self.min_r + 1 <= 32 and self.num_iter < 2**31
                      and (64 + 2 * (self.max_r + 2)) // self.grid + 2 <= 64)
        self.keys = torch.empty((P, self.num_iter), dtype=i32, device=dev) if self.keyed else None
        self.cap = max(1, min(self.num_iter, self.bitmap_words * 32))
        # keyed path: the unique keys per tile (arrival-ordered slices) and every tile's (first, count)
        self.unique_keys = torch.empty((P, self.cap), dtype=i32, device=dev) if self.keyed else None
        self.tile_ranges = torch.zeros((P, self.n_tiles, 2), dtype=i32, device=dev) if self.keyed else None
        self._tie_keys = None  # unique_keys while they describe self.circles (tie-breakers of the suppression)
        self.circles = torch.empty((P, self.cap, 3), dtype=i32, device=dev)
        self.scores = torch.empty((P, self.cap), dtype=torch.float32, device=dev)
        self.alive = torch.empty((P, self.cap), dtype=i32, device=dev)
        self.max_rc = torch.zeros((P, 2), dtype=i32, device=dev)
        self.state = torch.zeros((P, self.cap), dtype=u8, device=dev)
        per_rc, per_exp, per_starts = nat.perimeter_table(self.min_r, self.max_r)
        self.per_rc = torch.from_numpy(per_rc).to(dev)
        self.per_exp = torch.from_numpy(per_exp).to(dev)
        self.per_starts = torch.from_numpy(per_starts).to(dev)
        # keyed scoring (mg_score_circles_keyed: group-per-circle prefilter with orientation bounds, angles on
        # demand) whenever the radii fit it; else mg_score_circles, which reads the angle map.
        # MG_OLD_SCORE=1 forces the latter (A/B measurements, parity of the two paths)
        self.layer_starts = None
        self.keyed_score = bool(self.keyed and nat.lib().mg_score_keyed_supported(self.min_r, self.max_r) == 1
                                and not os.environ.get("MG_OLD_SCORE"))
        if self.keyed_score:
            self.pair_table = torch.from_numpy(nat.score_pair_table().view(np.int64)).to(dev)
            self.layer_starts = torch.zeros((P, self.n_tiles, self.max_r - self.min_r + 2), dtype=i32, device=dev)
            self.surv_list = torch.empty((P, self.cap, 2), dtype=i32, device=dev)  # (list index, key) per survivor
        self.nms_grid = None
        self.seeds = torch.zeros((P,), dtype=torch.int64, device=dev)
        self.seeds_host = torch.zeros((P,), dtype=torch.int64).pin_memory()
        self.raw = None
        self.stats = {}

    # -- host <-> device bookkeeping ---------------------------------------------------------------
    def _fetch_status(self, after=None):
        """The status block on the host (one pinned copy + a stream sync): rows as laid out in __init__.
        ``after`` (an event): the copy runs on a side stream that only waits for that event -- not for what the caller
        has queued behind it (the ROI pass of `follow`)."""
        if after is None:
            self.status_host.copy_(self.status, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return self.status_host.numpy()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
        self._side.wait_event(after)
        with torch.cuda.stream(self._side):
            self.status_host.copy_(self.status, non_blocking=True)
        self._side.synchronize()
        return self.status_host.numpy()

    @staticmethod
    def _hint(recent, floor, spare=1):
        """Launches to issue before the host looks: the most any of the last calls needed plus ``spare`` (a sweep /
        round after convergence costs a flag test), at least `floor`."""
        return max(floor, (max(recent) + spare) if recent else floor)

    @staticmethod
    def _note(recent, needed):
        recent.append(int(needed))
        del recent[:-4]

    # -- stage 1: to_uint8 + blur + Scharr/quantiles + Canny + hysteresis + edge grid --------
    def edge_stage(self, planes: torch.Tensor, minmax, low_q: float, high_q: float, keep_u8=False,
                   passthrough_u8=False, optimistic=False):
        """``optimistic``: nothing comes back to the host -- the sweeps and the coordinate capacity are those of the
        calls before and the caller (find) checks convergence / overflow when it fetches the status block."""
        L, P, h, w, s = nat.lib(), self.P, self.h, self.w, _stream()
        assert tuple(planes.shape) == (P, h, w)
        if planes.stride(2) != 1:
            planes = planes.contiguous()
        code = nat.dtype_code(planes.dtype)
        if passthrough_u8:
            if code != nat.MG_U8:
                raise TypeError("passthrough_u8 needs uint8 planes")
            minmax = None
        elif minmax is None:
            minmax = plane_minmax(planes)
        self.u8 = torch.empty((P, h, w), dtype=torch.uint8, device=self.dev) if keep_u8 else None
        # to_uint8 + blur and the one-pass combined histogram of m = dx^2 + dy^2 (exact below 8192, coarse (m >> 13)
        # above) of the blurred planes: one kernel for integer inputs, else two passes behind the same entry point
        self.hist.zero_()
        _call("mg_to_uint8_blur_hist", planes.data_ptr(), code, P, planes.stride(0), h, w, planes.stride(1), _ptr(minmax),
              self.blur.data_ptr(), _ptr(self.u8), self.hist.data_ptr(), self.hist_scratch.data_ptr(),
              self.hist_scratch.numel(), s)
        n = h * w
        idx = [quantile_indexes(n, q) for q in (low_q, high_q)]  # (prev, next, gamma) per quantile
        # Thresholds on the device (mg_edge_thresholds: no host round trip, no ATen kernels).  Ranks inside coarse bins
        # (strong gradients: noiseless images, the chip's chamber windows) take window passes that stay on the device as
        # well (_window_pass); as many are launched without looking as the calls before needed, and whether they
        # sufficed THIS time rides on the hysteresis convergence check below (status row `unresolved`).
        ranks4 = np.array([idx[0][0], idx[0][1], idx[1][0], idx[1][1]], dtype=np.int64)
        self._gammas = (float(idx[0][2]), float(idx[1][2]))
        _call("mg_edge_thresholds", self.hist.data_ptr(), P, ranks4.ctypes.data, self._gammas[0], self._gammas[1],
              self.thresh.data_ptr(), self.quant_d.data_ptr(), self.unresolved.data_ptr(), self.win_state.data_ptr(),
              self.hist_base.data_ptr(), s)
        self._quantiles = None
        n_win = max(self._recent_win) if self._recent_win else 0
        for k in range(n_win):
            self._window_pass(k)
        self.stats["hist_passes"] = 1 + n_win
        if optimistic:
            self._canny()
            self._sweeps(0, min(self._hint(self._recent_sweeps, 2), self.MAX_GROUP))
            self._edge_grid(3)
            n_edges = None
        else:
            n_edges, sweeps, needed, unresolved = self._edges_from_thresholds()
            if unresolved:  # more window passes than launched: one at a time until every plane is resolved, then again
                while unresolved:
                    self._window_pass(n_win)
                    n_win += 1
                    unresolved = bool((self._fetch_status()[2] & 0xFF).any())
                self.stats["hist_passes"] = 1 + n_win
                n_edges, sweeps, needed, _ = self._edges_from_thresholds()
            self._note(self._recent_win, int(self.status_host.numpy()[2].max()) >> 8)  # what the planes needed
            self.stats["hysteresis_sweeps"] = sweeps
            self._note(self._recent_sweeps, needed)
            need_cap = max(1, int(n_edges.max()))
            if self.coords is None or self.coords.shape[1] < need_cap:
                self.coords = torch.empty((P, int(need_cap * 1.25) + 1, 2), dtype=torch.int32, device=self.dev)
                self._drop_graphs()
            self._edge_grid(2)
        self.coord_cap = self.coords.shape[1]
        if self.keep_debug_maps:
            self.edges = torch.empty((P, h, w), dtype=torch.uint8, device=self.dev)
            _call("mg_unpack_bits", self.edge_bits.data_ptr(), self.words, P, h * w, self.edges.data_ptr(), s)
        if self.need_angle_map():
            if self.angle is None:
                self.angle = torch.empty((P, h, w), dtype=torch.float32, device=self.dev)
            _call("mg_edge_angles", self.blur.data_ptr(), P, h, w, self.coords.data_ptr(), self.coord_cap,
                  self.num_edges.data_ptr(), self.angle.data_ptr(), s)
        self.n_edges_host = n_edges
        return n_edges

    def _canny(self):
        _call("mg_canny_nms", self.blur.data_ptr(), self.P, self.h, self.w, self.thresh.data_ptr(), self.weak_bits.data_ptr(),
              self.edge_bits.data_ptr(), _ptr(self.class_bits), self.words, _stream())

    def _sweeps(self, done, group):
        """Hysteresis sweeps done .. done + group - 1; sweep k of the group counts its changes in self.changed[k]
        (cleared by the caller) and writes the tile flags of layer k + 1 (layer 0: the flags of the sweep before)."""
        flags = self.tile_flags
        if done > 0:
            flags[0].copy_(flags[self._last_layer])
        flags[1: group + 1].zero_()
        for g in range(group):
            _call("mg_canny_hysteresis", self.weak_bits.data_ptr(), self.edge_bits.data_ptr(), self.words, self.P, self.h,
                  self.w, self.changed[g].data_ptr(), flags[g].data_ptr() if done + g > 0 else 0,
                  flags[g + 1].data_ptr(), _stream())
        self._last_layer = group

    def _edge_grid(self, phases):
        coords = self.coords if phases & 2 else None
        _call("mg_edge_grid", self.edge_bits.data_ptr(), self.words, self.P, self.h, self.w, self.grid,
              self.cell_counts.data_ptr(), self.cell_starts.data_ptr(), self.num_edges.data_ptr(), _ptr(coords),
              coords.shape[1] if coords is not None else 0, self.scan_state.data_ptr(), self.edge_totals.data_ptr(), phases,
              _stream())

    def _window_pass(self, k):
        """Window pass k of the threshold search: the planes whose k-th distinct coarse bin holds one of their ranks
        re-histogram that bin exactly (base in self.hist_base, written by the kernel before; other planes are skipped)
        and resolve the ranks in it; a plane's last pass writes its thresholds."""
        if k > 3:
            raise RuntimeError("edge thresholds: more than four window passes")  # four ranks = at most four bins
        P = self.P
        if self.hist_win is None:
            self.hist_win = torch.zeros((P, FINE_BINS), dtype=torch.int32, device=self.dev)
            self._drop_graphs()
        else:
            self.hist_win.zero_()
        _call("mg_scharr_hist", self.blur.data_ptr(), P, self.h, self.w, 1, self.hist_base.data_ptr(), self.hist_win.data_ptr(),
              self.hist_scratch.data_ptr(), self.hist_scratch.numel(), _stream())
        _call("mg_edge_thresholds_window", self.hist_win.data_ptr(), P, k, self._gammas[0], self._gammas[1],
              self.win_state.data_ptr(), self.hist_base.data_ptr(), self.thresh.data_ptr(), self.quant_d.data_ptr(),
              self.unresolved.data_ptr(), _stream(), stage="mg_edge_thresholds")

    def _edges_from_thresholds(self):
        """Canny NMS + hysteresis to convergence + the cell counts of the edge grid, from self.thresh.
        Returns (n_edges per plane, sweeps run, sweeps needed, any plane whose thresholds still need window passes)."""
        self._canny()
        sweeps, unresolved, needed = 0, False, 0
        # sweeps per host check: first as many as the calls before needed (a sweep after convergence only
        # runs the tile-flag test), then two at a time -- one host round trip in the steady state
        group = min(self._hint(self._recent_sweeps, 2), self.MAX_GROUP)
        while True:
            self.changed[:group].zero_()
            self._sweeps(sweeps, group)
            sweeps += group
            # the cell counts of the edge grid ride on the same host round trip as the convergence check
            # (they are recomputed in the rare case that more sweeps are needed)
            self._edge_grid(1)
            st = self._fetch_status()
            per_sweep, n_edges = st[8: 8 + group].sum(axis=1), st[0].copy()
            unresolved = bool((st[2] & 0xFF).any())
            if unresolved:  # these edges come from invalid thresholds: the caller redoes them
                needed = sweeps
                break
            if per_sweep[group - 1] == 0:
                # sweeps this image needed = up to and including the first one that changed nothing
                needed = sweeps - group + int(np.argmax(per_sweep == 0)) + 1
                break
            group = 2
        return n_edges, sweeps, needed, unresolved

    @property
    def quantiles(self):
        """The two edge quantiles of every plane (float32, what np.quantile returns): (P, 2) on the host."""
        if self._quantiles is None:
            self._quantiles = self.quant_d.cpu().numpy()
        return self._quantiles

    def need_angle_map(self):
        """The dense angle map (mg_edge_angles) is only written for the scoring path that reads it or for the
        tests (keep_debug_maps); the keyed scoring computes the angles of its few exact sums on demand."""
        return self.keep_debug_maps or not self.keyed_score

    # -- stage 2: candidates -> unique integer circles -> scores ------------------------------
    def circle_stage(self, seeds, min_roundness: float, keep_raw=False, dedup_centres=False, counters_clear=False):
        """``counters_clear``: the caller has just cleared the whole status block (find's optimistic chain)."""
        L, P, h, w, s = nat.lib(), self.P, self.h, self.w, _stream()
        self.seeds_host.numpy()[:] = np.asarray(seeds, dtype=np.uint64).reshape(P).view(np.int64)
        self.seeds.copy_(self.seeds_host, non_blocking=True)  # pinned: queued on the stream, the host does not wait
        self.raw = torch.empty((P, self.num_iter, 3), dtype=torch.float32, device=self.dev) if keep_raw else None
        if self.keyed:
            _call("mg_candidate_keys", self.coords.data_ptr(), self.coord_cap, self.cell_starts.data_ptr(),
                  self.cell_counts.data_ptr(), self.num_edges.data_ptr(), P, h, w, self.grid, self.seeds.data_ptr(),
                  self.num_iter, self.min_r, self.max_r, self.keys.data_ptr(), _ptr(self.raw), s,
                  stage="mg_candidate_circles")
            _call("mg_keys_to_circles", self.keys.data_ptr(), self.num_iter, self.cell_starts.data_ptr(),
                  self.cell_counts.data_ptr(), self.num_edges.data_ptr(), P, h, w, self.grid, self.min_r, self.max_r,
                  self.unique_keys.data_ptr(), self.cap, self.tile_ranges.data_ptr(), self.num_circles.data_ptr(),
                  _ptr(self.layer_starts), int(counters_clear), s, stage="mg_bitmap_to_circles")
            self._tie_keys = self.unique_keys
        else:
            if self.bitmap is None:
                self.bitmap = torch.zeros((P, self.bitmap_words), dtype=torch.int32, device=self.dev)
                self.layer_offsets = torch.zeros((P, self.n_layers + 1), dtype=torch.int32, device=self.dev)
            self._tie_keys = None
            _call("mg_candidate_circles", self.coords.data_ptr(), self.coord_cap, self.cell_starts.data_ptr(),
                  self.cell_counts.data_ptr(), self.num_edges.data_ptr(), P, h, w, self.grid,
                  self.seeds.data_ptr(), self.num_iter, self.min_r, self.max_r,
                  self.bitmap.data_ptr(), self.bitmap_words, _ptr(self.raw), s)
            _call("mg_bitmap_to_circles", self.bitmap.data_ptr(), self.bitmap_words, P, h, w, self.min_r, self.max_r,
                  self.layer_offsets.data_ptr(), self.circles.data_ptr(), self.cap,
                  self.num_circles.data_ptr(), s)
        if not counters_clear:
            self.status[3:7].zero_()  # num_alive, num_out, num_scored, num_surv
        self.max_rc.fill_(-(2**31))
        if self.keyed_score and self._tie_keys is not None:
            _call("mg_score_circles_keyed", self.blur.data_ptr(), 0, self.edge_bits.data_ptr(), self.class_bits.data_ptr(),
                  self.words, P, h, w, self.circles.data_ptr(), self.cap, self.unique_keys.data_ptr(),
                  self.layer_starts.data_ptr(), self.min_r, self.max_r, self.per_rc.data_ptr(), self.per_exp.data_ptr(),
                  self.per_starts.data_ptr(), int(self.per_rc.shape[0]), self.pair_table.data_ptr(), float(min_roundness),
                  int(self.keep_debug_maps), self.scores.data_ptr(), self.alive.data_ptr(), self.num_alive.data_ptr(),
                  self.max_rc.data_ptr(), self.num_scored.data_ptr(), self.surv_list.data_ptr(), self.cap,
                  self.num_surv.data_ptr(), 1, s, stage="mg_score_circles")  # (num_surv: cleared above or by the caller)
            return
        if self.angle is None:  # a path switch after the edge stage (tests flip `keyed`): the map is needed after all
            self.angle = torch.empty((P, h, w), dtype=torch.float32, device=self.dev)
            _call("mg_edge_angles", self.blur.data_ptr(), P, h, w, self.coords.data_ptr(), self.coord_cap,
                  self.num_edges.data_ptr(), self.angle.data_ptr(), s)
        _call("mg_score_circles", self.angle.data_ptr(), self.edge_bits.data_ptr(), _ptr(self.class_bits), self.words, P, h, w,
              self.circles.data_ptr(), self.cap, _ptr(self.layer_offsets), _ptr(self._tie_keys),
              _ptr(self.tile_ranges if self._tie_keys is not None else None), self.min_r, self.max_r,
              self.per_rc.data_ptr(), self.per_exp.data_ptr(), self.per_starts.data_ptr(), int(self.per_rc.shape[0]),
              float(min_roundness), int(self.keep_debug_maps), int(dedup_centres), self.scores.data_ptr(),
              self.alive.data_ptr(),
              self.num_alive.data_ptr(), self.max_rc.data_ptr(), self.num_scored.data_ptr(), s)

    # -- stage 3: greedy suppression + ordered output -------------------------------------------
    def _out_buffers(self, need):
        """Ordered-output buffers (circles, scores, scratch) with room for ``need`` circles per plane; two sets used in
        turn, so the tables a caller got from the previous ``find`` stay intact during this one."""
        if need > self._out_cap:
            self._out_cap = int(need * 1.25) + 16
            self._out_sets = [None, None]
            self._drop_graphs()  # (captured for the old sets: their addresses may be handed out again)
        self._out_turn ^= 1
        if self._out_sets[self._out_turn] is None:
            P, cap = self.P, self._out_cap
            self._out_sets[self._out_turn] = (torch.empty((P, cap, 3), dtype=torch.int32, device=self.dev),
                                              torch.empty((P, cap), dtype=torch.float32, device=self.dev),
                                              torch.empty((P, 3 * cap), dtype=torch.int32, device=self.dev))
        return self._out_sets[self._out_turn]

    def _drop_graphs(self):
        """Forget every captured chain: a buffer its launches have baked in was made anew."""
        if self._graphs:
            if os.environ.get("MG_GRAPH_DESTROY"):
                self._graphs.clear()
            else:
                # the forgotten graphs are kept alive until the finder goes (a handful over a finder's life: buffers
                # regrow rarely): destroying a hipGraphExec between a replay and the next capture aborted the process
                # on ROCm 7.2
                self._retired_graphs.extend(self._graphs.values())
                self._graphs.clear()
        self._graph_seen.clear()

    def _collect(self, bufs, min_dist, cleared=False):
        """``cleared``: the status block (num_out) has been cleared since the last gather (the optimistic chain)."""
        self._n_collects += 1
        out, out_scores, scratch = bufs
        _call("mg_collect_circles", self.circles.data_ptr(), self.cap, self.scores.data_ptr(), self.alive.data_ptr(),
              self.num_alive.data_ptr(), self.state.data_ptr(), int(min_dist <= 0), self.P, out.data_ptr(),
              out_scores.data_ptr(), out.shape[1], self.num_out.data_ptr(), scratch.data_ptr(), _ptr(self._tie_keys),
              int(cleared), _stream())
        # what fetch_results' side stream waits for: the tables, not whatever the caller queues behind them
        if not torch.cuda.is_current_stream_capturing():  # (a graph replay records its own, _optimistic_chain)
            self._results_ready = torch.cuda.Event()
            self._results_ready.record()

    def _nms_prepare(self, min_dist):
        pad = 2 * min_dist + 1
        grid_cap = (self.h + self.max_r + 2 * pad) * (self.w + self.max_r + 2 * pad)
        if self.nms_grid is None or self.nms_grid.shape[1] < grid_cap:
            self.nms_grid = torch.empty((self.P, grid_cap), dtype=torch.int64, device=self.dev)
            self.nms_grid.fill_(-1)  # once: every call restores the cells it touched (mg_nms_cleanup)
            self.state.zero_()
            self._drop_graphs()
        if getattr(self, "_nms_dist", None) != min_dist:
            ring = nat.circle_points(min_dist, True)
            self._nms_ring = torch.from_numpy(ring).to(self.dev)
            self._nms_dbits = torch.from_numpy(_ring_difference_bits(ring, min_dist)).to(self.dev)
            if getattr(self, "nms_done", None) is None:
                self.nms_done = torch.zeros((self.P,), dtype=torch.int32, device=self.dev)
            self._nms_dist = min_dist
            self._drop_graphs()

    def _sparse_nms(self):
        """mg_nms_sparse ahead of the claim-grid rounds: from four planes of 512^2 on.  It runs a plane in ONE workgroup
        (~60 us of latencies whatever the plane holds) and the rounds' launches still go out, to find the planes done:
        a single plane (C1, C2, a one-timepoint shard) or hundreds of chamber windows with a few circles each (C3) are
        quicker through the rounds alone (C1: 0.87 against 0.96 ms)."""
        return _NMS_SPARSE and self.P >= 4 and self.h * self.w >= (1 << 18)

    def _nms_rounds(self, min_dist, first, count, out_cap, cleared=False):
        """Suppression rounds; round k of this group counts what it left undecided in self.undecided[k] (``cleared``:
        the status block they are rows of has just been cleared).  ``first``: the same-centre pass (only the first
        circle of a centre enters the rounds: exact, three tiny launches)."""
        P, s, ring = self.P, _stream(), self._nms_ring
        skip = self.nms_done.data_ptr() if self._sparse_nms() else 0
        if first:
            if skip:  # whole planes decided from the circles alone; the calls below leave those planes alone
                _call("mg_nms_sparse", self.circles.data_ptr(), self.cap, self.scores.data_ptr(), self.alive.data_ptr(),
                      self.num_alive.data_ptr(), self.max_rc.data_ptr(), P, min_dist, self._nms_dbits.data_ptr(),
                      self.state.data_ptr(), _ptr(self._tie_keys), self.nms_done.data_ptr(), s, stage="mg_nms_rounds")
            _call("mg_nms_same_centre", self.circles.data_ptr(), self.cap, self.scores.data_ptr(), self.alive.data_ptr(),
                  self.num_alive.data_ptr(), self.max_rc.data_ptr(), P, min_dist, self.nms_grid.data_ptr(),
                  self.nms_grid.shape[1], self.state.data_ptr(), _ptr(self._tie_keys), out_cap, skip, s, stage="mg_nms_rounds")
        if count > 0:
            _call("mg_nms_rounds", self.circles.data_ptr(), self.cap, self.scores.data_ptr(),
                  self.alive.data_ptr(), self.num_alive.data_ptr(), self.max_rc.data_ptr(), P,
                  min_dist, ring.data_ptr(), ring.shape[0], self.nms_grid.data_ptr(),
                  self.nms_grid.shape[1], self.state.data_ptr(), self.undecided.data_ptr(), self.undecided.stride(0),
                  int(count), int(cleared), _ptr(self._tie_keys), out_cap, skip, s, stage="mg_nms_rounds")

    def _nms_cleanup(self, min_dist, out_cap):
        _call("mg_nms_cleanup", self.circles.data_ptr(), self.cap, self.scores.data_ptr(), self.alive.data_ptr(),
              self.num_alive.data_ptr(), self.max_rc.data_ptr(), self.P, min_dist, self._nms_ring.data_ptr(),
              self._nms_ring.shape[0], self.nms_grid.data_ptr(), self.nms_grid.shape[1], self.state.data_ptr(),
              out_cap, self.nms_done.data_ptr() if self._sparse_nms() else 0, _stream())

    def nms_stage(self, min_dist: int, optimistic=False, bufs=None, cleared=False):
        """Checked chain: the alive counts come to the host first (they size the output), the rounds are checked for
        convergence afterwards.  ``optimistic``: output capacity and rounds from the calls before; the caller's (find's)
        single status fetch follows -- returns (buffers, rounds launched) and leaves the checks to ``_nms_finish``."""
        if optimistic:
            if bufs is None:
                bufs = self._out_buffers(self._out_cap)
            rounds = 0
            if min_dist > 0:
                self._nms_prepare(min_dist)
                # (a missing round is cheap to add -- unless a follow-up pass has been queued on the tables: _round_spare)
                rounds = min(self._hint(self._recent_rounds, 2, spare=self._round_spare), self.MAX_GROUP)
                self._nms_rounds(min_dist, True, rounds, bufs[0].shape[1], cleared=cleared)
            self._collect(bufs, min_dist, cleared=cleared)
            return bufs, rounds
        n_alive = self._fetch_status()[3]
        bufs = self._out_buffers(max(1, int(n_alive.max())))
        return self._nms_finish(min_dist, bufs, 0, int(n_alive.max()), None)

    def _nms_finish(self, min_dist, bufs, rounds, max_alive, st):
        """Rounds until nothing is undecided (``rounds`` were launched already and ``st`` is the status fetched after
        them; none: start here), the ordered output, the cleanup of the claim grid."""
        out_cap = bufs[0].shape[1]
        launched = rounds
        if min_dist > 0 and max_alive > 0:
            self._nms_prepare(min_dist)
            group = rounds
            while True:
                if st is not None:
                    per_round = st[8 + self.MAX_GROUP: 8 + self.MAX_GROUP + group].sum(axis=1)
                    if per_round[group - 1] == 0:
                        self._note(self._recent_rounds, rounds - group + int(np.argmax(per_round == 0)) + 1)
                        break
                    if rounds > 10000:
                        raise RuntimeError("greedy suppression did not converge")
                group = min(self._hint(self._recent_rounds, 2, spare=0), self.MAX_GROUP) if rounds == 0 else 2
                self._nms_rounds(min_dist, rounds == 0, group, out_cap)
                rounds += group
                # the ordered output is gathered before the check and rides on the same round trip (it is gathered
                # again in the rare case that more rounds are needed)
                self._collect(bufs, min_dist)
                st = self._fetch_status()
        elif st is None:
            self._collect(bufs, min_dist)
            st = self._fetch_status()
        counts = st[4].copy()
        self.stats["nms_rounds"] = rounds
        if rounds or launched:
            self._nms_cleanup(min_dist, out_cap)
        self._out_counts = counts.astype(np.int64)
        return bufs[0], bufs[1], self.num_out

    # -- the optimistic chain, eager or as a hipGraph -----------------------------------------------------------
    def _launch_chain(self, planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw, keep_u8, passthrough_u8,
                      bufs=None):
        """Everything between the inputs and the status fetch, launched without looking: returns (output buffers,
        suppression rounds launched).  ``bufs``: the ordered output goes there instead of into the next public set."""
        self.status.zero_()
        self.edge_stage(planes, minmax, low_q, high_q, keep_u8=keep_u8, passthrough_u8=passthrough_u8, optimistic=True)
        self.circle_stage(seeds, min_roundness, keep_raw=keep_raw, dedup_centres=min_dist > 0, counters_clear=True)
        return self.nms_stage(min_dist, optimistic=True, bufs=bufs, cleared=True)  # (status.zero_() above)

    def _optimistic_chain(self, planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw, keep_u8,
                          passthrough_u8, stable_input=False):
        """The chain as ONE hipGraph launch: the ~40 kernel launches / clears of a call are a fixed sequence as long as
        the inputs sit at the same addresses and the hints (window passes, sweeps, rounds, list and output capacities)
        have not moved, for each of the two output sets used in turn -- all of that is the graph's key (up to 16 graphs are
        kept).  What changes from call to call
        travels through fixed buffers: the per-plane min / max (copied into the finder's own block before the launch),
        the seeds (the pinned block the captured upload reads) and, for small inputs (<= 32 MB: the chip's 784 chamber windows, gathered afresh every call), the
        planes themselves (copied into the finder's own input block).  A sequence is captured the first time it shows
        when the input is the finder's block or the caller vouches for its buffer (``stable_input``), else the second
        time (a capture costs ~9 ms: not to be spent on addresses that never come back).  A single plane from a caller
        that does not vouch for its buffer is launched eagerly (measured through mg.beads: C2 1.06 ms eager, 1.07 replayed,
        C3 3.83 / 3.90;
        a one-timepoint StackProcessor shard, which does: 1.31 eager, 1.17 replayed).  Per-stage timing (a
        StageTimer without allow_graphs), debug maps and the raw / uint8 side outputs need the eager launches.
        MG_NO_GRAPH=1 turns the graphs off."""
        usable = (self._graphs is not None and (self.P >= 2 or stable_input) and (TIMER is _NO_TIMER or getattr(TIMER, "allow_graphs", False))
                  and not (keep_raw or keep_u8 or self.keep_debug_maps)
                  and not self.need_angle_map() and planes.stride(2) == 1)
        if not usable:
            if minmax is not None:
                minmax = minmax.contiguous()
            return self._launch_chain(planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw, keep_u8,
                                      passthrough_u8)
        if minmax is None and not passthrough_u8:
            minmax = plane_minmax(planes)
        if minmax is not None:
            # (stream-ordered: the values are read by the launch behind it; a strided view -- one channel's rows of a
            # (T, C, 2) block -- is gathered by this one copy)
            self._mm.copy_(minmax.reshape(self.P, 2))
            minmax = self._mm
        if not stable_input and planes.numel() * planes.element_size() <= (32 << 20):
            if self._in_stage is None or self._in_stage.dtype != planes.dtype:
                self._in_stage = torch.empty((self.P, self.h, self.w), dtype=planes.dtype, device=self.dev)
            self._in_stage.copy_(planes)
            planes, stable_input = self._in_stage, True
        win = max(self._recent_win) if self._recent_win else 0
        sweeps = min(self._hint(self._recent_sweeps, 2), self.MAX_GROUP)
        rounds = min(self._hint(self._recent_rounds, 2, spare=self._round_spare), self.MAX_GROUP) if min_dist > 0 else 0
        # the ordered output goes straight into the public set whose turn it is (two sets used in turn: a graph per
        # set -- twice the captures, once; a set of the graph's own cost two copies per call)
        bufs = self._out_buffers(self._out_cap)
        ready = ((self.nms_grid is not None or min_dist <= 0) and (self.hist_win is not None or win == 0)
                 and (min_dist <= 0 or getattr(self, "_nms_dist", None) == min_dist))  # nothing left to allocate / upload
        # EVERY buffer whose address or size is baked into the captured kernel arguments and that may be made anew
        # between two calls belongs to the key (the caching allocator can hand a regrown output set the
        # address of a dropped one -- a graph captured for the old capacity would then gather with the old stride):
        # the three tensors of the output set and its capacity, the coordinate list, the claim grid and ring, the
        # window histogram; and the switches that select other kernels.
        key = (planes.data_ptr(), planes.stride(0), planes.stride(1), planes.dtype, passthrough_u8, float(low_q), float(high_q),
               float(min_roundness), int(min_dist), win, sweeps, rounds, self.coords.data_ptr(), self.coords.shape[1],
               bufs[0].data_ptr(), bufs[0].shape[1], bufs[1].data_ptr(), bufs[2].data_ptr(),
               self.nms_grid.data_ptr() if self.nms_grid is not None else 0,
               self.nms_grid.shape[1] if self.nms_grid is not None else 0,
               self._nms_ring.data_ptr() if min_dist > 0 and getattr(self, "_nms_ring", None) is not None else 0,
               self.hist_win.data_ptr() if self.hist_win is not None else 0, bool(self.keyed), bool(self.keyed_score))
        entry = self._graphs.get(key)
        if entry is None and ready and self.graph_captures < self.MAX_CAPTURES + self.calls["optimistic"] // 64:
            # (hints that drift -- one sweep more or less -- add a few graphs; a caller whose sequences never repeat stops
            # capturing after MAX_CAPTURES)
            seen = stable_input or key in self._graph_seen
            if len(self._graph_seen) >= 16:
                self._graph_seen.clear()
            self._graph_seen.add(key)
            if seen:
                entry = self._capture_chain(key, planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, passthrough_u8,
                                            bufs)
                if self._graphs is None:
                    entry = None
        if entry is None:
            return self._launch_chain(planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw, keep_u8,
                                      passthrough_u8, bufs=bufs)
        self.seeds_host.numpy()[:] = np.asarray(seeds, dtype=np.uint64).reshape(self.P).view(np.int64)
        entry["graph"].replay()
        # the host-side state the eager launches would have left behind
        for name, value in entry["state"].items():
            setattr(self, name, value)
        self.stats.update(entry["stats"])
        self._n_collects += 1
        self._results_ready = torch.cuda.Event()
        self._results_ready.record()
        self.graph_replays += 1
        return bufs, entry["rounds"]

    def _capture_chain(self, key, planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, passthrough_u8, bufs):
        """Stream capture of _launch_chain (on a side stream; nothing runs); None if the capture fails -- the graphs
        of this finder are then off for good and the caller launches eagerly."""
        global _CAPTURING
        if len(self._graphs) >= 16:  # the oldest half goes (insertion order)
            for old_key in list(self._graphs)[:8]:
                del self._graphs[old_key]
        self.graph_captures += 1
        main = torch.cuda.current_stream()
        if getattr(self, "_cap_stream", None) is None:
            self._cap_stream = torch.cuda.Stream(device=self.dev)
        try:
            graph = torch.cuda.CUDAGraph()
            self._cap_stream.wait_stream(main)
            _CAPTURING = True
            # (thread-local: what other host threads do on their streams meanwhile does not concern this capture.
            # capture_begin / capture_end directly, not torch.cuda.graph: that context manager starts with
            # torch.cuda.empty_cache(), which hands every cached block back to the driver -- 0.4 s when a few tens of
            # gigabytes of grown-out-of output sets sit in the cache, inside whatever region the capture falls into)
            with torch.cuda.stream(self._cap_stream):
                graph.capture_begin(capture_error_mode="thread_local")
                try:
                    _, rounds = self._launch_chain(planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, False,
                                                   False, passthrough_u8, bufs=bufs)
                finally:
                    graph.capture_end()
            main.wait_stream(self._cap_stream)
        except Exception as exc:  # noqa: BLE001  (whatever the runtime refuses: fall back in-process)
            self._graphs = None
            self.stats["graph_error"] = f"{type(exc).__name__}: {exc}"
            try:  # a failed capture leaves its error behind for the next runtime call: take it here
                torch.cuda.synchronize(self.dev)
            except Exception:  # noqa: BLE001
                pass
            return None
        finally:
            _CAPTURING = False
        entry = {"graph": graph, "rounds": rounds,
                 "state": {k: getattr(self, k) for k in ("_last_layer", "coord_cap", "u8", "raw", "_tie_keys", "_quantiles",
                                                          "_gammas")},
                 "stats": {k: self.stats[k] for k in ("hist_passes",) if k in self.stats}}
        self._graphs[key] = entry
        return entry

    def find(self, planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw=False, keep_u8=False,
             passthrough_u8=False, host_results=True, follow=None, stable_input=False):
        """Returns per-plane lists (circles int32 (M,3) [row, col, r], scores float32 (M,)) on the
        host plus the device tensors (out, out_scores, num_out).  ``host_results=False``: only the
        counts come back -- (counts, (out, out_scores, num_out)); ``fetch_results`` copies the lists
        later (e.g. after the ROI pass, which reads the tables on the device, has been launched).
        ``out`` / ``out_scores`` stay valid until the next-but-one ``find`` of this finder (two buffer sets used in
        turn); ``num_out`` is a row of the status block and is cleared by the next ``find``.

        ``stable_input``: the caller hands in the same buffer call after call (StackProcessor's own image block): a
        hipGraph of the chain may be captured the first time a launch sequence shows (_optimistic_chain).

        ``follow(out, num_out, out_cap)``: device work that consumes the ordered tables (the ROI pass), queued BEFORE
        the host waits for the status block -- the GPU goes straight on instead of idling through the round trip and the
        launch after it.  It reads the tables and the per-plane counts where they are, on the device; if the optimistic
        chain has to be repaired (or the output gathered again) it is simply called again on the final tables.  Its last
        return value is ``self.follow_result``.

        Host round trips: the checked chain has three (hysteresis convergence + edge counts, alive counts,
        suppression convergence + output counts).  Once a call has gone through it, the next ones run OPTIMISTICALLY:
        sweeps, rounds and buffer capacities as the calls before needed (plus one), every kernel launched without
        looking, ONE fetch of the status block at the end -- which shows whether the thresholds were resolved, the
        sweeps converged, the coordinate list and the output were large enough and the rounds converged.  Anything
        else is repaired from that point with the checked chain (the results are the same either way: every stage is
        a pure function of its inputs; a plane whose edges overflowed the list was skipped by the kernels after it)."""
        opt = (self.optimistic and self.coords is not None and self._out_cap > 0
               and bool(self._recent_sweeps) and (min_dist <= 0 or bool(self._recent_rounds)))
        self.stats["optimistic"] = False
        # a suppression round too few is added after the status fetch at the price of a round trip; with a follow-up
        # pass queued on the tables it would also cost that pass again (the ROI pass: 4.5 ms at 64 planes, seen once in
        # 20 steps) -- one spare round (~10 us when nothing is undecided) is launched instead
        self._round_spare = 1 if follow is not None else 0
        if opt:
            bufs, rounds = self._optimistic_chain(planes, minmax, low_q, high_q, min_roundness, min_dist, seeds, keep_raw,
                                                  keep_u8, passthrough_u8, stable_input)
            followed = None
            if follow is not None:
                chain_done = torch.cuda.Event()
                chain_done.record()
                self.follow_result = follow(bufs[0], self.num_out, bufs[0].shape[1])
                followed = (self._n_collects, bufs[0].data_ptr())
                st = self._fetch_status(after=chain_done)  # the host looks at the counters while the ROI pass runs
            else:
                st = self._fetch_status()
            sweeps = self._last_layer
            per_sweep = st[8: 8 + sweeps].sum(axis=1)
            edges_ok = (not (st[2] & 0xFF).any() and per_sweep[sweeps - 1] == 0
                        and int(st[1].max()) <= self.coords.shape[1])
            if edges_ok:
                n_edges = st[0].copy()
                self.n_edges_host = n_edges
                self.stats.update(hysteresis_sweeps=sweeps, optimistic=True)
                self._note(self._recent_sweeps, int(np.argmax(per_sweep == 0)) + 1)
                self._note(self._recent_win, int(st[2].max()) >> 8)
                max_alive = int(st[3].max())
                if max_alive > bufs[0].shape[1]:  # the ordered output did not fit: gather it again (the rounds hold)
                    bufs = self._out_buffers(max_alive)
                    self._collect(bufs, min_dist)
                    st = self._fetch_status()
                out, out_scores, num_out = self._nms_finish(min_dist, bufs, rounds, max_alive, st)
            else:
                # the edges were not final (or did not fit): restore the claim grid the rounds wrote into, then redo
                self.calls["repaired"] += 1
                if rounds:
                    self._nms_cleanup(min_dist, bufs[0].shape[1])
                if int(st[1].max()) > self.coords.shape[1]:
                    self.coords = None  # sized again from the counts
                opt = False
        self.calls["optimistic" if opt else "checked"] += 1
        if not opt:
            if minmax is not None:
                minmax = minmax.contiguous()
            n_edges = self.edge_stage(planes, minmax, low_q, high_q, keep_u8=keep_u8, passthrough_u8=passthrough_u8)
            # with suppression to follow, passing circles that share a centre are reduced to their first
            self.circle_stage(seeds, min_roundness, keep_raw=keep_raw, dedup_centres=min_dist > 0)
            out, out_scores, num_out = self.nms_stage(min_dist)
        counts = self._out_counts  # came back with the suppression's convergence check
        self.stats["n_edges"] = n_edges
        if follow is not None and (not opt or followed != (self._n_collects, out.data_ptr())):
            self.calls["followed_again"] += int(opt)
            self.follow_result = follow(out, num_out, out.shape[1])  # (the tables it read before were not the final ones)
        if not host_results:
            return counts, (out, out_scores, num_out)
        return self.fetch_results(counts, out, out_scores), (out, out_scores, num_out)

    def fetch_results(self, counts, out, out_scores, overlap=False):
        """Host copies of the ordered circle lists.  ``overlap``: on a side stream that only waits for
        the suppression's outputs, so the copies run beside whatever was launched after ``find``."""
        # only the filled prefix comes back (the full-capacity copies left the GPU idle for ~0.5 ms)
        mx = max(int(counts.max()) if self.P else 0, 1)
        if overlap:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self.dev)
            self._side.wait_event(self._results_ready)
            with torch.cuda.stream(self._side):
                out.record_stream(self._side)
                out_scores.record_stream(self._side)
                out_h, sc_h = out[:, :mx].contiguous().cpu().numpy(), out_scores[:, :mx].contiguous().cpu().numpy()
        else:
            out_h, sc_h = out[:, :mx].contiguous().cpu().numpy(), out_scores[:, :mx].contiguous().cpu().numpy()
        return [(out_h[p, : counts[p]].copy(), sc_h[p, : counts[p]].copy()) for p in range(self.P)]


# --------------------------------------------------------------------------------------
# A12-A15, A18: labels, ROI gather, masks, reductions
# --------------------------------------------------------------------------------------

_HALF_CACHE = {}
_POOL = {}


def pooled(name, count, tail, dtype, device):
    """A (count, *tail) view of a grow-only device buffer.  Output sizes depend on the number of
    markers found, which varies from call to call; re-using one buffer (grown by 25 % when needed)
    keeps hipMalloc out of the steady state.  The view is valid until the next call with ``name``."""
    key = (name, tuple(tail), dtype, str(device))
    buf = _POOL.get(key)
    if buf is None or buf.shape[0] < count:
        _POOL[key] = None  # release the old block before growing
        buf = torch.empty((int(count * 1.25) + 1,) + tuple(tail), dtype=dtype, device=device)
        _POOL[key] = buf
    return buf[:count]


_ROI_POOL_NAMES = ("roi", "fg", "bg", "sums", "counts", "roi_offsets", "roi_order")


def drop_pool_tags(tags):
    """Free the ROI output sets pooled under these tags ("" = the untagged set)."""
    names = {n + t for n in _ROI_POOL_NAMES for t in tags}
    for key in [k for k in _POOL if k[0] in names]:
        del _POOL[key]


def release_pool():
    _POOL.clear()
    _BOUND_OF.clear()  # (the bounds lived in pool blocks)



def _halfwidth_table(max_r: int, device):
    key = (max_r, str(device))
    if key not in _HALF_CACHE:
        tab = np.full((max_r + 1, 2 * max_r + 1), -1, dtype=np.int32)
        for r in range(2, max_r + 1):
            tab[r, : 2 * r + 1] = nat.disk_halfwidths(r)
        _HALF_CACHE[key] = torch.from_numpy(tab).to(device)
    return _HALF_CACHE[key]


_LABEL_POOL = {}


def circle_labels(beads_per_assay, h, w, device="cuda", reuse=False):
    """utils.circle_labels (utils.py:380-395) for a list of (M_a, 3) int bead arrays.
    Returns labels (A, h, w) int32 on the device.  ``reuse=True`` hands out a pooled map that must
    be given back with ``release_labels`` (which restores -1 under the disks instead of clearing
    the whole map)."""
    require_gpu()
    a = len(beads_per_assay)
    cap = max(1, max((len(b) for b in beads_per_assay), default=1))
    host = np.zeros((a, cap, 3), dtype=np.int32)
    counts = np.zeros(a, dtype=np.int32)
    max_r = 2
    for k, b in enumerate(beads_per_assay):
        b = np.asarray(b).reshape(-1, 3)
        host[k, : len(b)] = b
        counts[k] = len(b)
        if len(b):
            max_r = max(max_r, int(b[:, 2].max()))
    key = (a, h, w, str(device))
    if reuse and key in _LABEL_POOL:
        labels = _LABEL_POOL.pop(key)
    else:
        labels = torch.full((a, h, w), -1, dtype=torch.int32, device=device)
    d_beads = torch.from_numpy(host).to(device)
    d_counts = torch.from_numpy(counts).to(device)
    tab = _halfwidth_table(max_r, device)
    _call("mg_circle_labels", d_beads.data_ptr(), cap, d_counts.data_ptr(), a, h, w, tab.data_ptr(), max_r,
          labels.data_ptr(), 0, _stream())
    labels._mg_state = (d_beads, cap, d_counts, a, h, w, tab, max_r, key)
    return labels


def release_labels(labels):
    """Give a label map back to the pool: -1 is restored under the disks that were drawn."""
    d_beads, cap, d_counts, a, h, w, tab, max_r, key = labels._mg_state
    _call("mg_circle_labels", d_beads.data_ptr(), cap, d_counts.data_ptr(), a, h, w, tab.data_ptr(), max_r,
          labels.data_ptr(), 1, _stream(), stage="mg_circle_labels_reset")
    _LABEL_POOL[key] = labels


_STAGING = {}


def _upload_i32(values, device):
    """A small int32 table on the device through pinned staging (queued on the stream: a pageable source makes
    the copy synchronous).  Three staging buffers per length are used in turn; a caller synchronises with the
    device at least once per hot-path step, long before a buffer comes round again."""
    n = len(values)
    ring = _STAGING.get((n, str(device)))
    if ring is None:
        ring = _STAGING[(n, str(device))] = [[torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(3)], 0]
    ring[1] = (ring[1] + 1) % 3
    host = ring[0][ring[1]]
    host.numpy()[:] = values
    return host.to(device, non_blocking=True)


_NMS_SPARSE = os.environ.get("MG_NMS_SPARSE", "1") != "0"


def _ring_difference_bits(ring, d):
    """Bitmap of D = ring (-) ring for mg_nms_sparse: bit (dr + 2 d)(4 d + 1) + dc + 2 d is set iff two rings of radius d
    whose centres differ by (dr, dc) have a cell in common (utils.py:266-289 marks and tests exactly those cells)."""
    ring = np.asarray(ring, dtype=np.int64)
    side = 4 * d + 1
    diff = (ring[:, None, :] - ring[None, :, :]).reshape(-1, 2) + 2 * d
    flat = np.unique(diff[:, 0] * side + diff[:, 1])
    bits = np.zeros(((side * side + 31) // 32,), dtype=np.uint32)
    np.bitwise_or.at(bits, flat >> 5, np.uint32(1) << (flat & 31).astype(np.uint32))
    return bits.view(np.int32)


_ROI_ORDER = os.environ.get("MG_ROI_ORDER", "1") != "0"
_ROI_ORDER_MIN = int(os.environ.get("MG_ROI_ORDER_MIN", "200000"))  # windows x planes: below, the pass is bound by latencies,
# not by its lines (8 timepoints of C4, 61 k: -0.02 ms for a 0.025 ms kernel; mode R, 1 932 markers x 256 planes: -0.45 ms)


def _window_order(d_beads, bead_stride, d_off, n_assays, m, pool_tag, planes=1):
    """The order mg_roi_segment_reduce visits the markers in (mg_roi_window_order: band by band, left to right), or
    None (MG_ROI_ORDER=0: as listed)."""
    if not _ROI_ORDER or m <= 0 or m * planes < _ROI_ORDER_MIN:
        return None
    dev = d_beads.device
    d_order = (pooled("roi_order" + pool_tag, m, (), torch.int32, dev) if pool_tag is not None
               else torch.empty((m,), dtype=torch.int32, device=dev))
    _call("mg_roi_window_order", d_beads.data_ptr(), int(bead_stride), d_off.data_ptr(), n_assays, m, d_order.data_ptr(), _stream())
    return d_order


def roi_gather_reduce(images: torch.Tensor, centers_per_assay, roi_len: int, labels: torch.Tensor | None,
                      want_roi=True, want_masks=True, want_sums=True, reuse_buffers=False, disks=False,
                      device_tables=None, time_major=False, device_counts=None, pool_tag=""):
    """images (A, C, T, h, w) -- or, with ``time_major`` (and ``disks``), (A, T, C, h, w): the outputs
    are (channel, time)-ordered either way; centers_per_assay: list of (M_a, >=2) int arrays [row, col, ...].

    Masks: from the ``labels`` map (utils.circle_labels) or, with ``disks=True`` and (M_a, 3) bead
    tables [row, col, r], straight from the bead geometry (same result, no label map at all).

    ``device_tables=(d_beads (A, cap, 3) int32, counts, max_r)`` (with ``disks``): the bead tables
    are still on the device, one padded row per assay as mg_collect_circles writes them; only the
    per-assay counts (host) are needed to size the outputs, ``centers_per_assay`` is ignored.

    ``device_counts=(d_counts (A,) int32 on the device, cap, bound)`` (with ``device_tables=(d_beads, None, max_r)``):
    the per-assay counts have not reached the host yet -- the pass is queued all the same, for at most ``bound``
    markers in all (None: A x cap) with the assays' offsets computed on the device (mg_counts_to_offsets, counts cut to
    ``cap``); ``finish_roi(res, counts)`` cuts the outputs to size once the counts are known and tells whether the
    bound held.

    Returns dict: roi (M, C, T, L, L), fg/bg (M, L, L) uint8, sums (M, C, T, 2) float64
    [fg sum, bg sum], counts (M, 2) int32, offsets (A+1,) numpy.  ``reuse_buffers`` returns views of
    pooled buffers that the next call overwrites (steady-state streaming use); ``pool_tag`` names a separate set of them
    (a streaming caller alternates two: the host copy of one chunk's outputs runs beside the next chunk's pass)."""
    require_gpu()
    if time_major:
        if not disks:
            raise ValueError("time_major needs the bead-table masks (disks=True)")
        a, t, c, h, w = images.shape
    else:
        a, c, t, h, w = images.shape
    images = images.contiguous()
    dev = images.device
    if device_counts is not None:
        d_counts, cap, bound = device_counts
        assert device_tables is not None and disks and d_counts.dtype == torch.int32 and d_counts.numel() == a
        sizes, m, offsets = None, (a * int(cap) if bound is None else max(1, min(int(bound), a * int(cap)))), None
    else:
        sizes = [int(n) for n in device_tables[1]] if device_tables is not None else [len(b) for b in centers_per_assay]
        m = int(sum(sizes))
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    res = {"offsets": offsets, "bound": m}
    L = int(roi_len)
    alloc = ((lambda name, n, tail, dt, d: pooled(name + pool_tag, n, tail, dt, d)) if reuse_buffers
             else (lambda name, n, tail, dt, d: torch.empty((n,) + tuple(tail), dtype=dt, device=d)))
    res["roi"] = alloc("roi", m, (c, t, L, L), images.dtype, dev) if want_roi else None
    res["fg"] = alloc("fg", m, (L, L), torch.uint8, dev) if want_masks else None
    res["bg"] = alloc("bg", m, (L, L), torch.uint8, dev) if want_masks else None
    res["sums"] = alloc("sums", m, (c, t, 2), torch.float64, dev) if want_sums else None
    res["counts"] = alloc("counts", m, (2,), torch.int32, dev) if want_sums else None
    if m == 0:
        return res
    if device_tables is not None:
        d_tab, _, max_r = device_tables
        assert disks and d_tab.dtype == torch.int32 and d_tab.is_contiguous() and d_tab.shape[0] == a
        max_r = max(int(max_r), 2)
        tab = _halfwidth_table(max_r, dev)
        if device_counts is not None:
            d_off = pooled("roi_offsets" + pool_tag, a + 1, (), torch.int32, dev)
            _call("mg_counts_to_offsets", d_counts.data_ptr(), a, min(int(cap), d_tab.shape[1]), d_off.data_ptr(), _stream())
        else:
            d_off = _upload_i32(offsets, dev)
        # where the markers' rows and offsets are on the device (marker_table reads them there)
        res["device_tables"] = (d_tab, int(d_tab.shape[1]), d_off)
        d_order = _window_order(d_tab, d_tab.shape[1], d_off, a, m, pool_tag if reuse_buffers else None, c * t)
        _call("mg_roi_segment_reduce", images.data_ptr(), nat.dtype_code(images.dtype), c * t * h * w, c, t, h, w,
              int(time_major), d_tab.data_ptr(), d_tab.shape[1], d_off.data_ptr(), a, m, _ptr(d_order), L, tab.data_ptr(), max_r,
              _ptr(res["roi"]), _ptr(res["fg"]), _ptr(res["bg"]), _ptr(res["sums"]), _ptr(res["counts"]), _stream())
        return res
    beads = np.zeros((m, 3), dtype=np.int32)
    assay = np.zeros(m, dtype=np.int32)
    local = np.zeros(m, dtype=np.int32)
    for k, b in enumerate(centers_per_assay):
        b = np.asarray(b)
        lo, hi = offsets[k], offsets[k + 1]
        if hi > lo:
            beads[lo:hi, : 3 if disks else 2] = b[:, : 3 if disks else 2]
            assay[lo:hi] = k
            local[lo:hi] = np.arange(hi - lo)
    d_beads = torch.from_numpy(beads).to(dev)
    if disks:
        max_r = max(int(beads[:, 2].max()), 2)
        tab = _halfwidth_table(max_r, dev)
        d_off = torch.from_numpy(offsets.astype(np.int32)).to(dev)
        d_order = _window_order(d_beads, 0, d_off, a, m, pool_tag if reuse_buffers else None, c * t)
        _call("mg_roi_segment_reduce", images.data_ptr(), nat.dtype_code(images.dtype), c * t * h * w, c, t, h, w,
              int(time_major), d_beads.data_ptr(), 0, d_off.data_ptr(), a, m, _ptr(d_order), L, tab.data_ptr(), max_r,
              _ptr(res["roi"]), _ptr(res["fg"]), _ptr(res["bg"]), _ptr(res["sums"]), _ptr(res["counts"]), _stream())
        return res
    d_assay = torch.from_numpy(assay).to(dev)
    d_local = torch.from_numpy(local).to(dev)
    _call("mg_roi_gather_reduce_batched", 
        images.data_ptr(), nat.dtype_code(images.dtype), c * t * h * w, c, t, h, w, d_beads.data_ptr(),
        d_assay.data_ptr(), d_local.data_ptr(), m, L, _ptr(labels), _ptr(res["roi"]), _ptr(res["fg"]),
        _ptr(res["bg"]), _ptr(res["sums"]), _ptr(res["counts"]), _stream())
    return res


def finish_roi(res, counts):
    """The outputs of a ``roi_gather_reduce(device_counts=...)`` call cut to the markers there are, now that the
    per-assay counts are on the host (the kernel placed them compactly, assay after assay).  None if there are more
    markers than the pass was launched for (the caller runs it again with the counts it now has)."""
    counts = np.asarray(counts, dtype=np.int64)
    m = int(counts.sum())
    if m > res["bound"]:
        return None
    out = {k: (v[:m] if isinstance(v, torch.Tensor) else v) for k, v in res.items()}
    out["offsets"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return out


def marker_table(out: dict, assay_offset: int, n_channels: int) -> torch.Tensor:
    """Rows [assay, row, col, r, fg_count, bg_count, fg_sum[C], bg_sum[C]] (float64: exact for these integers) of a
    ``roi_gather_reduce`` / ``StackProcessor`` result, assembled on the device by one kernel (mg_marker_table) from the
    bead tables where they are; a result that only has host bead lists uploads them first."""
    require_gpu()
    offsets = out.get("offsets")
    if offsets is None:
        offsets = np.concatenate([[0], np.cumsum([len(b) for b in out["beads"]])])
    offsets = np.asarray(offsets, dtype=np.int64)
    m, a = int(offsets[-1]), len(offsets) - 1
    sums, counts = out["sums"], out["counts"]
    dev = sums.device
    tab = torch.empty((m, 6 + 2 * n_channels), dtype=torch.float64, device=dev)
    if m == 0:
        return tab
    assert sums.shape[1] == n_channels and sums.is_contiguous() and counts.is_contiguous()
    tables = out.get("device_tables")
    if tables is not None:
        d_beads, stride, d_off = tables
    else:
        beads = [np.asarray(b, dtype=np.int32).reshape(-1, 3) for b in out["beads"]]
        d_beads, stride, d_off = _upload_i32(np.concatenate(beads).reshape(-1), dev), 0, _upload_i32(offsets, dev)
    _call("mg_marker_table", d_beads.data_ptr(), stride, d_off.data_ptr(), a, m, int(assay_offset), counts.data_ptr(),
          sums.data_ptr(), n_channels, int(sums.shape[2]), 0, tab.data_ptr(), _stream())
    return tab


def masked_median(roi: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """nanmedian of roi (M, C, T, L, L) -- uint8, uint16, float32 or float64 -- under mask (M, L, L) (one mask for all
    timepoints) or (M, T, L, L) (a mask per timepoint; a broadcast view with time stride 0 is read in place)
    -> (M, C, T) float64 (identify.py:76-80, filter.py:20-22)."""
    require_gpu()
    m, c, t, L, _ = roi.shape
    roi = roi.contiguous()
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if mask.dim() == 3:
        mask = mask.contiguous()
        sm, st = L * L, 0
    else:
        if mask.shape[1] not in (1, t):
            raise ValueError(f"masked_median: {mask.shape[1]} mask timepoints for {t} roi timepoints")
        if mask.stride(3) != 1 or mask.stride(2) != L or (m > 1 and mask.stride(0) < L * L):
            mask = mask.contiguous()
        sm, st = (mask.stride(0) if m > 1 else L * L), (mask.stride(1) if mask.shape[1] == t and t > 1 else 0)
    out = torch.empty((m, c, t), dtype=torch.float64, device=roi.device)
    if m == 0:
        return out
    _call("mg_roi_masked_median", roi.data_ptr(), nat.dtype_code(roi.dtype), mask.data_ptr(), sm, st, m, c, t, L,
          out.data_ptr(), _stream())
    return out


def masked_median_u16(roi: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """nanmedian of roi (M, C, T, L, L) uint16 under mask (M, L, L) -> (M, C, T) float64."""
    require_gpu()
    m, c, t, L, _ = roi.shape
    out = torch.empty((m, c, t), dtype=torch.float64, device=roi.device)
    _call("mg_roi_masked_median_u16", roi.contiguous().data_ptr(), mask.contiguous().data_ptr(), m, c, t, L,
          out.data_ptr(), _stream())
    return out


# --------------------------------------------------------------------------------------
# A16 / A17: ButtonFinder helpers
# --------------------------------------------------------------------------------------

_CV_CACHE = {}


def _cv_table(max_r: int, device):
    """Rows r = 0..max_r of cv.circle's filled-disk half widths (utils.py:38 call site)."""
    key = (max_r, str(device))
    if key not in _CV_CACHE:
        tab = np.full((max_r + 1, max_r + 1), -1, dtype=np.int32)
        for r in range(max_r + 1):
            tab[r, : r + 1] = nat.cv_disk_halfwidths(r)
        _CV_CACHE[key] = torch.from_numpy(tab).to(device)
    return _CV_CACHE[key]


def cluster_1d(points, total_length, num_clusters, cluster_length, ideal_num_points, penalty, device="cuda"):
    """find.py:632-677: brute-force search over integer offsets (costs on the GPU, one thread per
    offset), first minimum wins; returns the cluster label of every point (-1 outside)."""
    require_gpu()
    points = np.asarray(points, dtype=np.float64)
    perm = np.argsort(points)
    pts = points[perm]
    n_off = int(total_length - round(num_clusters * cluster_length))
    if n_off <= 0:
        raise ValueError("cluster_1d: the clusters do not fit into total_length")  # reference: best_spans is None
    d_pts = torch.from_numpy(pts).to(device) if len(pts) else torch.zeros(1, dtype=torch.float64, device=device)
    d_ideal = torch.from_numpy(np.asarray(ideal_num_points, dtype=np.float64)).to(device)
    costs = torch.empty(n_off, dtype=torch.float64, device=device)
    _call("mg_cluster1d_costs", d_pts.data_ptr(), len(pts), n_off, int(num_clusters), float(cluster_length),
          d_ideal.data_ptr(), float(penalty), costs.data_ptr(), _stream())
    best = int(np.argmin(costs.cpu().numpy()))  # first occurrence == strict "<" update rule
    bounds = np.arange(num_clusters + 1) * cluster_length + best
    spans = np.searchsorted(pts, bounds)
    labels = -np.ones(len(pts), dtype=int)
    labels[spans[0] : spans[-1]] = np.repeat(np.arange(num_clusters), spans[1:] - spans[:-1])
    return labels[np.argsort(perm)]


def button_masks(centers_rc, radii, roi_len, outer_r, inner_r, device="cuda"):
    """fg = filled cv.circle(radius_i), bg = annulus(outer_r, inner_r) about centers (row, col) in
    window coordinates (find.py:383-400).  Returns uint8 tensors (M, L, L)."""
    require_gpu()
    centers_rc = np.ascontiguousarray(np.asarray(centers_rc, dtype=np.int32).reshape(-1, 2))
    radii = np.ascontiguousarray(np.asarray(radii, dtype=np.int32).reshape(-1))
    m = len(radii)
    max_r = int(max([outer_r, inner_r] + radii.tolist())) if m else max(outer_r, inner_r)
    tab = _cv_table(max_r, device)
    fg = torch.empty((m, roi_len, roi_len), dtype=torch.uint8, device=device)
    bg = torch.empty((m, roi_len, roi_len), dtype=torch.uint8, device=device)
    if m:
        d_c, d_r = torch.from_numpy(centers_rc).to(device), torch.from_numpy(radii).to(device)
        _call("mg_button_masks", d_c.data_ptr(), d_r.data_ptr(), m, int(roi_len), int(outer_r), int(inner_r),
              tab.data_ptr(), tab.shape[1], max_r, fg.data_ptr(), bg.data_ptr(), _stream())
    return fg, bg


def masked_sums(roi: torch.Tensor, fg: torch.Tensor, bg: torch.Tensor):
    """roi (M, C, T, L, L), fg/bg (M, L, L) uint8 -> sums (M, C, T, 2) float64, counts (M, 2) int32."""
    require_gpu()
    m, c, t, L, _ = roi.shape
    sums = torch.empty((m, c, t, 2), dtype=torch.float64, device=roi.device)
    counts = torch.empty((m, 2), dtype=torch.int32, device=roi.device)
    if m:
        _call("mg_masked_sums", roi.contiguous().data_ptr(), nat.dtype_code(roi.dtype), fg.contiguous().data_ptr(),
              bg.contiguous().data_ptr(), m, c * t, L, sums.data_ptr(), counts.data_ptr(), _stream())
    return sums, counts
