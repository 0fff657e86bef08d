"""Operator API of the MI355X ToF Gaussian rasterizer.

Host-side mirror of the reference's
``submodules/diff-gaussian-rasterization-w-tof/diff_gaussian_rasterization_w_tof/__init__.py``
(same names, argument meaning, return arity and error behaviour) so that
``gaussian_renderer/__init__.py`` and ``train.py`` run unchanged:

  * ``GaussianRasterizationSettings``  -- reference __init__.py:22-40
  * ``GaussianRasterizer``             -- reference __init__.py:208-269
  * ``_RasterizeGaussians`` (autograd) -- reference __init__.py:69-206

The native work is done by libgftorf_rast.so (hand-written gfx950 HIP kernels)
through the C ABI in include/gftorf_rast.h.  PyTorch is used for device memory,
streams and autograd plumbing only.  There is no CPU path: tensors must live on
a HIP device and the library must be built, otherwise this raises.
"""
import ctypes as C
import struct as _struct
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    near_n: Optional[float] = 0.01
    far_n: Optional[float] = 100.0
    depth_range: Optional[float] = 100.0
    use_view_dependent_phase: Optional[bool] = False
    optimize_phase_offset: Optional[bool] = False
    optimize_dc_offset: Optional[bool] = False


# (Gaussian, tile) instance count of the most recent forward (diagnostics / bench)
last_call_stats = {"num_rendered": 0, "binning_instances": 0, "restarted": False, "max_tile_list": 0,
                   "forwards": 0, "restarts": 0}

# Diagnostics (bench.py): with keep_last_buffers the scratch buffers of the most recent forward stay referenced here
# so that a caller can read e.g. the per-quadrant list depths (layout: _lib.get_layout).  Off by default.
keep_last_buffers = False
last_call_buffers = {}

import os as _os
from collections import OrderedDict as _OrderedDict


class _CameraSchedule:
    """What the device keeps for ONE camera (image size + address of its view matrix) from one frame to the next -- schedules,
    never results: any contents give the same outputs (tests/test_gpu_parity.py::test_tile_hints_do_not_change_results).

    tile_hints    one word per tile: did a quadrant walk past where a sorted list head ends (gft_forward_io.tile_hints)?  The next
                  forward sorts such a tile's whole list up front instead of head first / rest on demand.
    tile_weights  the quadrants' walk lengths: the next forward blend deals its waves heaviest tile first (gft_forward_io.tile_weights).
    cell_sched    where the (supertile, slab) lists of the next frame start and how much they hold: no count pass from the
                  camera's second frame on (gft_forward_io.cell_sched); `sched_seen`: a counted frame has written it;
                  `miss_streak` / `sched_off_until`: after four frames in a row whose lists did not fit, the camera counts
                  again for 64 frames, then tries the schedule anew.
    hinted_tiles  how many tiles the schedule marked when the last frame read it (gft_forward_report.hinted_tiles): from a
                  sixteenth of the tiles on the next forward runs the whole-list build of the pull kernel.
    A camera is meant to be rendered on ONE stream at a time: the forward's closing workgroup rewrites the words in place, so
    two forwards of the same camera running concurrently on two streams would bin by each other's half-written schedule
    (harmless for values, a guaranteed schedule miss for time).  The 256 most recently used cameras are kept; an evicted
    camera's buffers are freed by torch in stream order behind the forwards that used them."""
    __slots__ = ("tile_hints", "tile_weights", "cell_sched", "sched_seen", "miss_streak", "sched_off_until", "hinted_tiles", "frames")

    def __init__(self):
        self.tile_hints = self.tile_weights = self.cell_sched = None
        self.sched_seen = False
        self.miss_streak = self.sched_off_until = self.hinted_tiles = self.frames = 0


class _OperatorState:
    """Everything the operator keeps between calls, in one place (`gftorf_amd.api.state`; `state.reset()` drops all of it --
    tests, or a caller that switches scenes).  Sizes and schedules only, never a result:

    instance_hint  (device, P, W, H[, slot]) -> (instance count, longest tile list) of the recent forwards: the binning buffer
                   is sized before the device has counted (gft_forward: no host round trip in the middle of the forward); a frame
                   that needs more re-runs stage 2 with the exact size.  The 64 most recent shapes.
    cameras        tiles_key -> _CameraSchedule, least recently used first, at most MAX_CAMERAS.
    status         hint key -> status block of the no-host-read flow (device words + pinned copy), the 64 most recent shapes.
    grad_pool, acc_pool   gradient tensors / accumulators kept from one backward to the next (see below).
    bw_plans, geom_bytes, image_bytes   host-side caches of shapes and sizes."""
    MAX_CAMERAS = 256
    MAX_SHAPES = 64

    def __init__(self):
        self.instance_hint = {}
        self.cameras = _OrderedDict()
        self.status = _OrderedDict()
        self.grad_pool = {}
        self.acc_pool = {}
        self.bw_plans = {}
        self.geom_bytes, self.image_bytes = {}, {}

    def reset(self):
        for d in (self.instance_hint, self.cameras, self.status, self.grad_pool, self.acc_pool, self.bw_plans, self.geom_bytes,
                  self.image_bytes):
            d.clear()

    def reset_schedules(self):
        """Forget every camera's schedules (the size hints stay)."""
        self.cameras.clear()

    def camera(self, key, create=True):
        cam = self.cameras.get(key)
        if cam is None:
            if not create:
                return None
            cam = self.cameras[key] = _CameraSchedule()
            while len(self.cameras) > self.MAX_CAMERAS:
                self.cameras.popitem(last=False)
        else:
            self.cameras.move_to_end(key)
        return cam


state = _OperatorState()
_instance_hint = state.instance_hint          # (the same objects under the names the tests and bench.py have always used)
# GFT_BWD_DETERMINISTIC=1: the backward forms its per-Gaussian sums in a fixed order instead of with float atomics
# (bit-reproducible gradients; several times slower: for tests)
_DETERMINISTIC = _os.environ.get("GFT_BWD_DETERMINISTIC", "0") != "0"
_HINT_HEADROOM = 1.25
# Schedule switches (all on; each one off gives the same results slower: INTEGRATION.md section J)
_TILE_HINTS = _os.environ.get("GFT_TILE_HINTS", "1") != "0"
_TILE_HINTS_PER_CAMERA = _os.environ.get("GFT_TILE_HINTS_PER_CAMERA", "1") != "0"
_CELL_SCHED = _os.environ.get("GFT_CELL_SCHED", "1") != "0"
_FWD_ORDER = _os.environ.get("GFT_FWD_ORDER", "1") != "0"
_force_cell_sched = None       # tests: True = bin by whatever the words hold
_WHOLE_SHARE = 16
_force_whole_lists = None      # tests: True / False overrides the choice of the build
_SCHED_MISSES_OFF = 4          # frames in a row whose lists did not fit the camera's schedule before it counts again ...
_SCHED_OFF_FRAMES = 64         # ... for this many frames


def _whole_lists(cam, n_tiles):
    if _force_whole_lists is not None:
        return int(bool(_force_whole_lists))
    return int((cam.hinted_tiles if cam is not None else 0) * _WHOLE_SHARE >= n_tiles)


# The forward without a host read (gft_forward_enqueue): taken automatically while the current stream is being captured into
# a graph (torch.cuda.graphs around a fixed-shape iteration: at the reference's own scene size the host, not the kernels,
# bounds the loop), and for every call with `gftorf_amd.api.no_host_read = True` / GFT_NO_HOST_READ=1.  The binning buffer is
# sized from the shape's earlier frames as always; a frame whose instance count exceeds it is NOT re-rendered (nobody reads
# the count while it runs): its outputs are undefined, and a later call of the shape -- which finds the device's count of
# overflowed frames in pinned memory: a word the library never clears, so a host that runs several frames ahead cannot miss
# it -- raises.  `enqueue_status()` returns the postings (after a graph replay, say).  Outside a capture, a shape that has
# no size hint yet (the first frame of the process, a new P after densification) takes the blocking two-stage flow once.
no_host_read = _os.environ.get("GFT_NO_HOST_READ", "0") != "0"
_status = state.status    # hint key -> dict(dev=int32[16] on the device, host=its pinned copy, seen=overflows reported so far)
_ST_CAP, _ST_OVERFLOWS, _ST_MAX_R = 12, 13, 14      # include/gftorf_rast.h: GFT_STATUS_CAP / _OVERFLOWS / _MAX_R


def _status_of(key, dev, create):
    st = _status.get(key)
    if st is None and create:
        st = _status[key] = dict(dev=torch.zeros((16,), device=dev, dtype=torch.int32),
                                 host=torch.zeros((16,), dtype=torch.int32).pin_memory(), seen=0, key=key)
        st["np"] = st["host"].numpy()
        while len(_status) > state.MAX_SHAPES:       # (P changes with every densification step: the most recent shapes)
            _status.popitem(last=False)
    elif st is not None:
        _status.move_to_end(key)
    return st


def enqueue_status(synchronize=True):
    """What the device posted for the no-host-read forwards of every shape: a list of dicts(key=(device, P, W, H[, slot]),
    posted, num_rendered, binning_instances -- of the most recent frame, both from the device's posting --, overflow,
    overflows, max_overflow_instances).  ``overflow``: the most recent frame did not fit its binning buffer and its outputs
    are undefined -- render it again eagerly (the blocking flow re-sizes the buffer); ``overflows``: how many frames of the
    shape did not, since the process started (a count the device keeps and nothing clears: frames whose posting a later
    frame has overwritten are in it)."""
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
        for st in _status.values():
            st["host"].copy_(st["dev"])
    out = []
    for key, st in _status.items():
        a = st["np"]
        out.append(dict(key=key, posted=bool(a[3]), num_rendered=int(a[0]), binning_instances=int(a[_ST_CAP]),
                        overflow=bool(a[3]) and int(a[0]) > int(a[_ST_CAP]), overflows=int(a[_ST_OVERFLOWS]),
                        max_overflow_instances=int(a[_ST_MAX_R]), prefiltered_point_culled=bool(a[1] & 1)))
    return out


# Gradient tensors kept from one backward to the next.  The operator returns dense gradient tensors (376 B per Gaussian
# with SH colour + SH phasor of 16 coefficients) of which a dense frame fills a few per cent of the rows (the Gaussians
# some pixel blended); writing the zeros of all the other rows is most of the backward's preprocess kernel (65 of 100 us at
# 1 M Gaussians).  A set of gradient tensors that nobody references any more -- the optimizer has consumed them and
# `zero_grad(set_to_none=True)` or the next backward has dropped them -- is therefore kept: the next backward zeroes the rows
# the last one wrote (they are marked in `dirty`) and writes only the rows of its own blended Gaussians
# (cfg.grads_zeroed = 3).  Reuse is decided per call and only when it is provably safe:
#   * no tensor aliases the buffer any more (storage use count back at its baseline; torch._C._storage_Use_Count, the
#     private counter CUDA-graph trees use -- without it nothing is reused), and
#   * nobody wrote to it through a tensor (version counter unchanged: an in-place op on `p.grad`, e.g. gradient
#     accumulation over two backwards or clipping, bumps it; the library's kernels write through raw pointers).
# Otherwise a fresh buffer is taken and written in full, exactly as before.  GFT_GRADS_REUSE=0 switches it off.
# CONTRACT of the reuse: gradients handed out by the operator are written through tensors only (autograd,
# optimizers, clipping, `p.grad.add_()`: all bump the version counter) -- a write through `.data`, `.detach()`-free raw
# pointers or another extension is invisible here and would leave non-zero rows the next backward does not know about.
# GFT_GRADS_REUSE_CHECK=1 verifies before every reuse that the unmarked rows are still zero (a debug mode: it reads the
# whole buffer) and raises if not.
_GRADS_REUSE = _os.environ.get("GFT_GRADS_REUSE", "1") != "0"
_GRADS_CHECK = _os.environ.get("GFT_GRADS_REUSE_CHECK", "0") != "0"
# "Nobody aliases the buffer any more" is read from the storage's use count where torch has the (private) counter
# CUDA-graph trees use.  Without it -- another torch version; GFT_GRADS_LIFETIME=dlpack forces it -- the public route:
# the pool owns the memory and never hands it out; every forward gets an ALIAS of it through DLPack (torch.from_dlpack of a
# capsule made here), whose deleter torch calls when the last tensor on that alias dies.  The version counter of an alias
# dies with it, so on this route in-place writes are not seen afterwards; the one writer that is not the caller's doing --
# autograd summing a second gradient INTO a tensor it took over as a leaf's `.grad` -- is kept out by holding a reference
# to the handed-out tensors until the next forward of the shape: autograd then copies into `.grad` instead of taking them
# (a copy per directly fed leaf: the price of the fallback), everything else is the contract above.
_USE_COUNT_API = hasattr(torch._C, "_storage_Use_Count") and _os.environ.get("GFT_GRADS_LIFETIME", "") != "dlpack"
_grad_pool = state.grad_pool       # (device, P, layout) -> list of {buf, dirty, version, base}
_DENSE_SHARE = 0.3        # rows written by the last rows-only backward / P above which the tensors are written in full
_DENSE_RUN = 15           # ... for this many backwards, before a rows-only one counts again
_GRAD_POOL_DEPTH = 3      # rasterizer calls of one iteration whose gradient tensors are alive at the same time


_report_slab = []         # [pinned int32 array, next slot]


def _report_slot():
    """One pinned host word for a pool entry's row count (gft_backward_io.rows_report), as a 1-element view of a process-wide
    array that is never freed: the backward that stores into it may still be in flight when its entry is dropped from the
    pool (P changes with every densification), and a freed pinned block may be handed to another `pin_memory()` user.
    Slots are reused round robin after 4096 entries; a stale store then lands in another entry's count -- a schedule
    (rows-only or full write), never a value."""
    if not _report_slab:
        _report_slab.extend([torch.zeros((4096,), dtype=torch.int32).pin_memory(), 0])
    i = _report_slab[1]
    _report_slab[1] = (i + 1) % 4096
    slot = _report_slab[0][i:i + 1]
    slot.zero_()
    return slot


def _storage_refs(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DL_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DL_DELETER)]
_dl_live = {}             # token -> (struct, shape array, callback, entry): alive until torch has called the deleter
_dl_token = [0]


def _dl_alias(entry):
    """A float32 tensor on the memory of ``entry["mem"]`` with a storage of its own (DLPack, kDLROCM): when the last tensor
    on that storage dies torch calls the capsule's deleter, which marks the entry free."""
    mem = entry["mem"]
    _dl_token[0] += 1
    token = _dl_token[0]
    m = _DLManagedTensor()
    shape = (C.c_int64 * 1)(mem.numel())
    m.dl_tensor.data = mem.data_ptr()
    m.dl_tensor.device = _DLDevice(10 if mem.is_cuda else 1, mem.device.index or 0)      # kDLROCM / kDLCPU
    m.dl_tensor.ndim = 1
    m.dl_tensor.dtype = _DLDataType(2, 32, 1)                                               # kDLFloat, 32 bits
    m.dl_tensor.shape = shape
    m.dl_tensor.strides = None
    m.dl_tensor.byte_offset = 0

    def released(_ptr, token=token, entry=entry):
        entry["free"] = True
        _dl_live.pop(token, None)
    cb = _DL_DELETER(released)
    m.deleter = cb
    _dl_live[token] = (m, shape, cb, entry)
    entry["free"] = False
    new = C.pythonapi.PyCapsule_New
    new.restype, new.argtypes = C.py_object, [C.c_void_p, C.c_char_p, C.c_void_p]
    return torch.from_dlpack(new(C.addressof(m), b"dltensor", None))


# The backward's accumulator (64 B per Gaussian) kept from one backward to the next.  The render backward adds to the rows
# of the Gaussians some pixel blended and the preprocess backward reads each of those rows once; with cfg.acc_zeroed = 2
# it zeroes them behind the read, so the buffer is all zero again after every backward and the next forward has nothing
# to clear (64 MB of HBM writes per frame at 1 M Gaussians, issued beside the binning kernels: 14-27 us of a 0.42 ms
# step).  A buffer is internal -- no tensor of it is ever handed out --, is reused only on the stream its last kernels ran
# on (or where the caller orders the streams itself, gftorf_amd.pair), and only if its last user left it zero: a forward
# whose backward never ran (its clear or its predecessor's zeroing stands), or a backward that returned without error.
# GFT_ACC_REUSE=0 switches it off (every forward then clears a fresh buffer, as before).
_ACC_REUSE = _os.environ.get("GFT_ACC_REUSE", "1") != "0"
_acc_pool = state.acc_pool         # (device, P) -> list of (buffer whose rows are zero, stream of its last kernels)
_ACC_POOL_DEPTH = 3


class _AccLease:
    """One accumulator buffer on its way through a forward and (maybe) a backward."""
    __slots__ = ("buf", "key", "stream", "zero", "was_zero", "pooled")

    def __init__(self, buf, key, stream, was_zero):
        self.buf, self.key, self.stream = buf, key, stream
        self.pooled = True
        self.was_zero = was_zero      # taken from the pool: the forward clears nothing
        self.zero = False             # the buffer is (in stream order) all zero and nobody is going to write to it

    def give_back(self):
        buf, self.buf = self.buf, None
        if buf is None or not self.zero or not _ACC_REUSE or not self.pooled:
            return
        pool = _acc_pool.setdefault(self.key, [])
        pool.append((buf, self.stream))
        del pool[:-_ACC_POOL_DEPTH]
        if len(_acc_pool) > 8:
            _acc_pool.pop(next(iter(_acc_pool)))

    def __del__(self):
        # a forward whose backward never ran: the accumulator was not touched after the forward's clear
        try:
            self.give_back()
        except Exception:
            pass


def _take_acc(lib, dev, P, stream, any_stream=False, pooled=True, prezero=False):
    key = (dev.index, P)
    pool = _acc_pool.get(key) if (_ACC_REUSE and pooled) else None
    if pool:
        for i, (buf, st) in enumerate(pool):
            if st == stream or any_stream:
                del pool[i]
                return _AccLease(buf, key, stream, True)
    if prezero:
        # (a forward that is queued without a host read may turn out not to fit its binning buffer: its kernels -- the one that
        # clears the accumulator as a side job among them -- then do nothing, and a buffer that counts as zero afterwards must be)
        lease = _AccLease(torch.zeros((lib.gft_acc_bytes(P) // 4,), device=dev, dtype=torch.float32), key, stream, True)
    else:
        lease = _AccLease(torch.empty((lib.gft_acc_bytes(P) // 4,), device=dev, dtype=torch.float32), key, stream, False)
    lease.pooled = pooled          # (a buffer allocated while a graph is captured belongs to the graph's pool: never kept)
    return lease
_LIST_HEADROOM = 1.2      # longest tile list of the previous frame -> guess for this one


_EMPTY = torch.Tensor([])      # "tensor absent" (reference __init__.py:239-252)


def _prod(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def _present(t):
    """The reference marks an absent tensor by an empty one (``torch.Tensor([])``)."""
    return t is not None and isinstance(t, torch.Tensor) and t.numel() != 0


def _f32(t, dev, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    if t.device != dev:
        raise RuntimeError("%s is on %s, expected %s" % (name, t.device, dev))
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _scalar(x):
    """phase_offset / dc_offset arrive as a Python float or a 1-element tensor
    (reference __init__.py:111-112; gaussian_renderer/__init__.py:126-127)."""
    if isinstance(x, torch.Tensor):
        return float(x.detach().reshape(-1)[0].item())
    return float(x)


def _bg_strides(bg, H, W, dev):
    """Background as [7,H,W] element strides.  The reference reads
    bg[c*H*W + pix] for c < 7 on every call (forward.cu:644,649) after
    ``.contiguous()``; an expanded constant background (train.py:127) is
    consumed through its zero strides instead of being materialised."""
    if bg.dtype != torch.float32 or bg.device != dev:
        raise RuntimeError("bg must be a float32 tensor on %s" % (dev,))
    if bg.dim() == 3 and tuple(bg.shape) == (7, H, W):
        return bg, bg.stride(0), bg.stride(1), bg.stride(2)
    if bg.numel() == 7 * H * W:
        bg = bg.contiguous()
        return bg, H * W, W, 1
    if bg.numel() == 7:
        bg = bg.contiguous().reshape(7)
        return bg, 1, 0, 0
    raise RuntimeError(
        "bg must hold 7*H*W floats ([7,H,W], may be an expanded view) or 7 floats; got shape %s. "
        "The ToF rasterizer blends 7 phasor planes against bg planes 0..6 on every call."
        % (tuple(bg.shape),))


_CFG_FMT = "=6i8f7i4x3q"       # gft_config field by field (include/gftorf_rast.h); checked against the ctypes struct below
assert _struct.calcsize(_CFG_FMT) == C.sizeof(_lib.Config)


def _make_config(s, P, M, M_p, H, W, phase_offset, dc_offset, bgs, want_backward):
    c = _lib.Config()
    # (one pack into the struct's buffer instead of twenty-four attribute stores: host time; acc_zeroed, grads_zeroed and
    # grads_accumulate start at 0)
    _struct.pack_into(_CFG_FMT, c, 0, P, int(s.sh_degree), M, M_p, W, H, float(s.tanfovx), float(s.tanfovy),
                      float(s.scale_modifier), float(s.near_n), float(s.far_n), float(s.depth_range), phase_offset, dc_offset,
                      1 if s.use_view_dependent_phase else 0, 1 if s.prefiltered else 0, 1 if s.debug else 0,
                      1 if want_backward else 0, 0, 0, 0, bgs[0], bgs[1], bgs[2])
    return c


def rasterize_gaussians(means3D, means2D, sh, sh_p, colors_precomp, phasors_precomp, opacities,
                        scales, rotations, cov3Ds_precomp, phase_offset, dc_offset, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, sh_p, colors_precomp, phasors_precomp,
                                     opacities, scales, rotations, cov3Ds_precomp, phase_offset,
                                     dc_offset, raster_settings)


def _canonical_cap(n):
    """Binning capacities are multiples of 64 instances: the buffer then holds exactly 12 bytes per instance plus
    one alignment unit, so the capacity can be read back from the buffer's size (``binning_capacity``) -- the
    pybind-level backward (``_C.rasterize_gaussians_backward``) gets the buffer, not the capacity."""
    return (int(n) + 63) // 64 * 64


_STABLE_CAPS = _os.environ.get("GFT_STABLE_CAPS", "1") != "0"
_POISON = _os.environ.get("GFT_POISON_SCRATCH", "0") != "0"


def _guess_cap(hint, sched_cells=0):
    """Capacity of the binning buffer for a frame whose instance count is guessed from recent frames: the guess plus 25 %,
    rounded UP to four significant bits (steps of at most 6 %).  Over changing views the guess moves a little every frame;
    without the rounding every frame asks the allocator for a buffer of another size -- a cache miss of torch's allocator
    whenever the size class changes (a guard: +-0 on a box with a fast host, 2364 -> 2408 it/s over 30 views).
    `sched_cells`: lists of the camera's list schedule -- every list reserves its last count plus a quarter plus 64 entries,
    and the schedule is only accepted if all of that fits the entry array: on a small scene (entries ~ instances) or a
    grid of many cells the 64 per list are more than the 4096 spare instances, and the camera would miss every frame."""
    n = int(hint * _HINT_HEADROOM) + 4096 + 64 * int(sched_cells)
    if _STABLE_CAPS and n > 16:
        sh = n.bit_length() - 4
        n = ((n + (1 << sh) - 1) >> sh) << sh
    return _canonical_cap(n)


def _scratch(nbytes, dev):
    t = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    if _POISON:
        t.fill_(0x7f)
    return t


def binning_capacity(binning, W, H):
    """Instances a binning buffer allocated by :func:`native_forward` for a W x H frame holds."""
    return int(_lib.load().gft_binning_capacity(binning.numel(), int(W), int(H)))


class _Settings(NamedTuple):
    """The scalar / camera arguments of ``_C.rasterize_gaussians`` (rasterize_points.h:25-53) by name."""
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    tanfovx: float
    tanfovy: float
    image_height: int
    image_width: int
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    near_n: float
    far_n: float
    depth_range: float
    use_view_dependent_phase: bool


_PLANE_SPLIT = (3, 7, 1, 3, 1, 1, 1, 1, 3)     # color, phasor, depth, normal, acc, entropy, depth_distortion, amp_distortion, distribution
_PLANE_FIRST = (0, 3, 10, 11, 14, 15, 16, 17, 18)   # first plane of each output inside the [21, H, W] allocation
_bw_plans = state.bw_plans
_geom_bytes, _image_bytes = state.geom_bytes, state.image_bytes      # gft_geom_bytes(P), gft_image_bytes(W, H): one library call per size
_FWD_FMT = "=%dQ" % len(_lib.ForwardIO._fields_)
_BWD_FMT = "=%dQ" % len(_lib.BackwardIO._fields_)
assert _struct.calcsize(_FWD_FMT) == C.sizeof(_lib.ForwardIO) and _struct.calcsize(_BWD_FMT) == C.sizeof(_lib.BackwardIO)
_FWD_AT = {n: i for i, (n, _t) in enumerate(_lib.ForwardIO._fields_)}
_BWD_AT = {n: i for i, (n, _t) in enumerate(_lib.BackwardIO._fields_)}


def _p0(t):
    return 0 if t is None else t.data_ptr()


def native_forward(s, means3D, sh, sh_p, colors_precomp, phasors_precomp, opacities, scales, rotations,
                   cov3Ds_precomp, ph_off, dc_off, want_bw, with_acc, stream=None, hint_slot=0, share_grads=None,
                   pre_launch=None, acc_any_stream=False, nowait=None):
    """One forward of the native rasterizer (``RasterizeGaussiansCUDA``, rasterize_points.cu:42-165): allocates the
    outputs and the three scratch buffers, runs the C ABI on torch's current stream.  ``s`` holds the settings fields
    (``GaussianRasterizationSettings`` or ``_Settings``), ``ph_off`` / ``dc_off`` are floats.  Returns a dict.

    For :mod:`gftorf_amd.pair`: ``stream`` = raw hipStream_t to launch on instead of torch's current stream (the caller
    orders it against the current stream), ``hint_slot`` keeps the buffer-size hints of the two cameras of a pair apart,
    ``share_grads`` = the ``prep`` of the pair's other view, whose gradient tensors this view's backward adds to,
    ``pre_launch`` = called after every host-side tensor preparation of this call (contiguous / aligned copies of the
    inputs and camera constants, queued on torch's CURRENT stream) and before its first kernel launch: a caller that
    launches on another stream orders that stream behind those copies there."""
    lib = _lib.load()
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("gftorf_amd: the rasterizer runs on a HIP device only (means3D is on %s); "
                           "there is no CPU path" % (dev,))
    # under graph capture nothing may be read from the device and no buffer of the eager pools may be baked into the graph
    capturing = torch.cuda.is_current_stream_capturing()
    if nowait is None:
        nowait = capturing or no_host_read
    if capturing and s.debug:
        raise RuntimeError("gftorf_amd: raster_settings.debug synchronises after every stage and cannot be captured in a graph")
    P = means3D.size(0)
    H, W = int(s.image_height), int(s.image_width)

    g = lambda t, n: _f32(t, dev, n) if _present(t) else None
    means3D_c = _f32(means3D, dev, "means3D") if P else means3D
    sh_c, sh_p_c = g(sh, "shs"), g(sh_p, "shs_p")
    colors_c, phasors_c = g(colors_precomp, "colors_precomp"), g(phasors_precomp, "phasors_precomp")
    opac_c = g(opacities, "opacities")
    scales_c, rot_c, cov_c = g(scales, "scales"), g(rotations, "rotations"), g(cov3Ds_precomp, "cov3D_precomp")
    view_c = _f32(s.viewmatrix, dev, "viewmatrix")
    proj_c = _f32(s.projmatrix, dev, "projmatrix")
    campos_c = _f32(s.campos, dev, "campos")
    bg_c, bsc, bsy, bsx = _bg_strides(s.bg, H, W, dev)
    M = sh_c.size(1) if sh_c is not None else 0
    M_p = sh_p_c.size(1) if sh_p_c is not None else 0

    if s.debug:
        cpu_args = cpu_deep_copy_tuple((s.bg, means3D, colors_precomp, phasors_precomp, opacities, scales,
                                        rotations, s.scale_modifier, cov3Ds_precomp, s.viewmatrix, s.projmatrix,
                                        s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh, sh_p,
                                        s.sh_degree, s.campos, s.prefiltered, s.debug, s.near_n, s.far_n,
                                        s.depth_range, s.use_view_dependent_phase, ph_off, dc_off))

    f32 = dict(device=dev, dtype=torch.float32)
    planes = torch.empty((21, H, W), **f32)
    # (one split instead of nine slices: host time)
    color, phasor, depth, normal, acc, entropy, depth_distortion, amp_distortion, distribution = planes.split(_PLANE_SPLIT)
    radii = torch.empty((P,), device=dev, dtype=torch.int32)
    pixels = torch.empty((P, 1), **f32)
    gb = _geom_bytes.get(P)
    if gb is None:
        gb = _geom_bytes[P] = int(lib.gft_geom_bytes(P))
    ib = _image_bytes.get((W, H))
    if ib is None:
        ib = _image_bytes[(W, H)] = int(lib.gft_image_bytes(W, H))
    geom = torch.empty((gb,), device=dev, dtype=torch.uint8)
    img = torch.empty((ib,), device=dev, dtype=torch.uint8)
    if _POISON:
        # debug (GFT_POISON_SCRATCH=1): the scratch buffers start as 0x7f bytes (3.4e38 as a float, 2139062143 as an index) instead
        # of whatever the allocator's block held -- usually the previous frame's plausible values --, so a kernel that reads a field
        # no kernel of this frame wrote shows up in the parity tests
        geom.fill_(0x7f)
        img.fill_(0x7f)

    cfg = _make_config(s, P, M, M_p, H, W, ph_off, dc_off, (bsc, bsy, bsx), want_bw)
    io = _lib.ForwardIO()
    # (the argument block in one pack -- field order of gft_forward_io --, the outputs' addresses from the one allocation
    # they are planes of: host time; binning, acc, grads_zero, tile_hints follow below)
    pl, hw4 = planes.data_ptr(), 4 * H * W
    _struct.pack_into(_FWD_FMT, io, 0,
                      _p0(bg_c), means3D_c.data_ptr() if P else 0, _p0(colors_c), _p0(phasors_c), _p0(opac_c), _p0(scales_c),
                      _p0(rot_c), _p0(cov_c), view_c.data_ptr(), proj_c.data_ptr(), campos_c.data_ptr(), _p0(sh_c), _p0(sh_p_c),
                      geom.data_ptr(), img.data_ptr(), 0,
                      pl, pl + 3 * hw4, pl + 10 * hw4, pl + 11 * hw4, pl + 14 * hw4, pl + 15 * hw4, pl + 16 * hw4, pl + 17 * hw4,
                      pixels.data_ptr() if P else 0, pl + 18 * hw4, radii.data_ptr() if P else 0, 0, 0, 0, 0, 0, 0)
    # the backward's accumulator: cleared by the forward beside its binning kernels -- unless it comes from the pool of
    # buffers that the last backward left zero (_AccLease)
    lease = None
    if want_bw and with_acc and P:
        lease = _take_acc(lib, dev, P, stream if stream is not None else _lib.raw_stream(dev), acc_any_stream, pooled=not capturing,
                          prezero=nowait and not capturing)
    acc_buf = lease.buf if lease is not None else None
    io.acc = None if (lease is not None and lease.was_zero) else _ptr(acc_buf)

    # the backward's gradient tensors and argument block, while the device is still busy with earlier work
    prep = None
    if want_bw and with_acc and P:
        prep = prepare_backward(s, means3D_c, opac_c, sh_c, sh_p_c, scales_c, rot_c, cov_c, radii, geom, img,
                                (bg_c, bsc, bsy, bsx), (view_c, proj_c, campos_c), ph_off, dc_off, acc_buf,
                                colors_c is not None, cov_c is not None, want_bw, pixels,
                                share_grads=share_grads, acc_lease=lease,
                                pooled=not capturing)
        if prep["zero_buf"] is not None:
            io.grads_zero = prep["zero_buf"].data_ptr()
            io.grads_zero_bytes = prep["zero_buf"].numel() * 4
    R = cap = 0
    restarted = False
    max_list = C.c_int64(0)
    if P == 0:
        # the reference skips every kernel and returns its zero-filled outputs
        # (rasterize_points.cu:104)
        planes.zero_()
        binning = torch.empty((0,), device=dev, dtype=torch.uint8)
    else:
        if stream is None:
            stream = _lib.raw_stream(dev)
        if pre_launch is not None:
            pre_launch()
        num_rendered = C.c_int64(0)
        hint_key = (dev.index, P, W, H) if not hint_slot else (dev.index, P, W, H, hint_slot)
        hint, list_hint = _instance_hint.get(hint_key, (None, 0))
        # (a schedule buffer made during a capture would live in the graph's private pool: only one that exists already)
        n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        # (the schedules are about regions of the image as ONE CAMERA sees them -- which tiles hold the scene's silhouette
        # differs from view to view: they are kept per image size and camera (_CameraSchedule), the camera being known by the
        # address of its view matrix (the reference's Camera objects keep theirs for the whole run, scene/cameras.py; a
        # caller that builds new matrices every call gets a fresh, empty schedule each time and evicts the oldest).  Not per
        # number of Gaussians: they survive the densification steps, which change P every hundred iterations.)
        tiles_key = (dev.index, W, H, hint_slot, s.viewmatrix.data_ptr() if _TILE_HINTS_PER_CAMERA else 0)
        # (a schedule buffer made during a capture would live in the graph's private pool: only what exists already)
        cam = state.camera(tiles_key, create=not capturing) if _TILE_HINTS else None
        use_sched = 0
        if cam is not None:
            cam.frames += 1
            if cam.tile_hints is None and not capturing:
                cam.tile_hints = torch.zeros((n_tiles,), device=dev, dtype=torch.int32)
            io.tile_hints = _ptr(cam.tile_hints)
            if cam.tile_hints is not None and _CELL_SCHED:
                if cam.cell_sched is None and not capturing:
                    words = int(lib.gft_cell_sched_words(W, H))
                    cam.cell_sched = torch.zeros((words,), device=dev, dtype=torch.int32) if words else False
                if cam.cell_sched is not None and cam.cell_sched is not False:
                    io.cell_sched = cam.cell_sched.data_ptr()
                    use_sched = int(_force_cell_sched if _force_cell_sched is not None
                                    else (cam.sched_seen and cam.frames >= cam.sched_off_until))
            if cam.tile_hints is not None and _FWD_ORDER:
                if cam.tile_weights is None and not capturing:
                    cam.tile_weights = torch.zeros((4 * n_tiles + 4,), device=dev, dtype=torch.int32)
                io.tile_weights = _ptr(cam.tile_weights)
        try:
            with _lib.on_device(dev):
                st = None
                if nowait:
                    st = _status_of(hint_key, dev, create=not capturing) if hint is not None else None
                    if capturing and hint is None:
                        raise RuntimeError("gftorf_amd: a forward captured in a graph sizes its binning buffer from earlier "
                                           "frames of the shape: render one frame of this shape (%d Gaussians, %dx%d) eagerly "
                                           "first" % (P, W, H))
                    if capturing and st is None:
                        raise RuntimeError("gftorf_amd: no status block for this shape: render one frame of it eagerly before "
                                           "capturing")
                    # (not capturing and nothing known about the shape yet -- the first frame of the process, a new P after a
                    # densification step --: the blocking two-stage flow below, once; it leaves the hint and the status block)
                if st is not None:
                    a = st["np"]
                    if int(a[_ST_OVERFLOWS]) != st["seen"]:
                        # some earlier no-host-read frame of the shape did not fit its buffer (the device counts them in a word
                        # nothing clears: the posting of the frame itself may have been overwritten by a later frame's already)
                        n_over, st["seen"] = int(a[_ST_OVERFLOWS]) - st["seen"], int(a[_ST_OVERFLOWS])
                        worst = int(a[_ST_MAX_R])
                        prev_r, prev_l = _instance_hint.get(hint_key, (0, 0))
                        _instance_hint[hint_key] = (max(worst, prev_r or 0), prev_l)
                        raise RuntimeError("gftorf_amd: %d earlier no-host-read forward(s) of this shape had up to %d instances, "
                                           "more than their binning buffer held: their outputs were undefined (the buffer has "
                                           "been enlarged for the following frames)" % (n_over, worst))
                    if a[3]:
                        # the posting of an earlier no-host-read frame of the shape (whichever the copy in pinned memory holds)
                        prev_R = int(a[0])
                        a[3] = 0
                        if cam is not None:
                            cam.hinted_tiles = int(a[8])
                        if a[1] & 1:
                            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
                        prev_r, prev_l = _instance_hint.get(hint_key, (0, 0))
                        _instance_hint[hint_key] = (max(prev_R, int((prev_r or 0) * 0.95)), prev_l)
                        hint = _instance_hint[hint_key][0]
                    cap = _guess_cap(hint)
                    binning = _scratch(lib.gft_binning_bytes(cap, W, H), dev)
                    io.binning = binning.data_ptr()
                    hints = _lib.ForwardHints(binning_instances=cap, max_tile_list=int(list_hint * _LIST_HEADROOM) + 1,
                                              whole_lists=_whole_lists(cam, n_tiles))
                    _lib.check(lib.gft_forward_enqueue(stream, C.byref(cfg), C.byref(io), C.byref(hints), st["dev"].data_ptr()))
                    st["host"].copy_(st["dev"], non_blocking=True)
                    R = -1                     # (not known to the host)
                elif hint is None:
                    # first frame of this shape: size the buffer after the one blocking
                    # read, like the reference's resize callback
                    # (rasterize_points.cu:27-33, rasterizer_impl.cu:311-315)
                    _status_of(hint_key, dev, create=True)      # (so that a later capture of this shape finds its status block)
                    _lib.check(lib.gft_forward_preprocess(stream, C.byref(cfg), C.byref(io),
                                                          C.byref(num_rendered), C.byref(max_list)))
                    R = int(num_rendered.value)
                    cap = _canonical_cap(R)
                    binning = _scratch(lib.gft_binning_bytes(cap, W, H), dev)
                    io.binning = binning.data_ptr()
                    _lib.check(lib.gft_forward_render(stream, C.byref(cfg), C.byref(io), cap, int(max_list.value)))
                else:
                    # later frames: the buffer is sized from the recent frames' instance counts (the only thing taken
                    # from earlier frames), both stages are queued back to back
                    cap = _guess_cap(hint, (cam.cell_sched.numel() - 4) // 2 if (cam is not None and use_sched) else 0)
                    binning = _scratch(lib.gft_binning_bytes(cap, W, H), dev)
                    io.binning = binning.data_ptr()
                    hints = _lib.ForwardHints(binning_instances=cap, max_tile_list=int(list_hint * _LIST_HEADROOM) + 1,
                                              whole_lists=_whole_lists(cam, n_tiles), use_cell_sched=use_sched)
                    report = _lib.ForwardReport()
                    _lib.check(lib.gft_forward(stream, C.byref(cfg), C.byref(io), C.byref(hints), C.byref(report)))
                    R = int(report.num_rendered)
                    if use_sched:
                        # (a camera whose lists keep outgrowing what its last frame left -- its tensors shared by scenes of
                        # different sizes, say -- pays the counted flow ON TOP of the failed attempt: after four misses in a row
                        # it counts again for 64 frames, then tries anew.  A frame that did not fit the BINNING BUFFER either is
                        # re-rendered below whatever the schedule said: not the schedule's miss.)
                        if report.sched_misses:
                            last_call_stats["sched_misses"] = last_call_stats.get("sched_misses", 0) + 1
                            if R <= cap:
                                cam.miss_streak += 1
                                if cam.miss_streak >= _SCHED_MISSES_OFF and _force_cell_sched is None:
                                    cam.miss_streak = 0
                                    cam.sched_off_until = cam.frames + _SCHED_OFF_FRAMES
                        else:
                            cam.miss_streak = 0
                    if cam is not None:
                        cam.hinted_tiles = int(report.hinted_tiles)
                    max_list.value = int(report.max_tile_list)
                    if R > cap:
                        restarted = True
                        cap = _canonical_cap(R)
                        binning = _scratch(lib.gft_binning_bytes(cap, W, H), dev)
                        io.binning = binning.data_ptr()
                        _lib.check(lib.gft_forward_render(stream, C.byref(cfg), C.byref(io), cap, int(max_list.value)))
                # slowly decaying maximum: alternating views of one scene (colour / ToF camera,
                # random training views) keep the larger count as the guess
                if st is None:
                    prev_r, prev_l = _instance_hint.get(hint_key, (0, 0))
                    _instance_hint[hint_key] = (max(R, int((prev_r or 0) * 0.95)), max(int(max_list.value), int(prev_l * 0.95)))
                    if len(_instance_hint) > state.MAX_SHAPES:
                        _instance_hint.pop(next(iter(_instance_hint)))
        except Exception as ex:
            if s.debug:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise ex
    if P and io.cell_sched:
        cam.sched_seen = True
    if prep is not None:
        prep["cap"] = cap
    if lease is not None:
        lease.zero = True          # (cleared by this forward, or zero since its last backward)

    last_call_stats.update(num_rendered=R, binning_instances=cap, restarted=restarted,
                           max_tile_list=int(max_list.value) if P else 0)
    if keep_last_buffers:
        last_call_buffers.update(geom=geom, img=img, binning=binning, P=P, W=W, H=H, cap=cap)
    last_call_stats["forwards"] = last_call_stats.get("forwards", 0) + 1
    last_call_stats["restarts"] = last_call_stats.get("restarts", 0) + int(restarted)
    return dict(R=R, cap=cap, outputs=(color, phasor, depth, normal, acc, entropy, depth_distortion, amp_distortion,
                                       pixels, distribution, radii),
                geom=geom, binning=binning, img=img, acc=acc_buf, prep=prep, bg=(bg_c, bsc, bsy, bsx),
                consts=(view_c, proj_c, campos_c),
                inputs=(means3D_c, opac_c, sh_c, sh_p_c, scales_c, rot_c, cov_c, colors_c, phasors_c))


def prepare_backward(s, means3D, opac, sh, sh_p, scales, rotations, cov3D, radii, geom, img, bg, consts, ph_off, dc_off,
                     acc, want_colors, want_cov, want_bw_records=True, pixels=None, zero_fill=False, share_grads=None,
                     acc_lease=None, pooled=True):
    """Everything of a backward that does not depend on the upstream gradients: the gradient tensors, the argument
    block, the config.  The forward calls it BEFORE it queues its kernels, so that this host work overlaps the device's
    previous work instead of sitting between the forward's last kernel and the backward's first one."""
    lib = _lib.load()
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(s.image_height), int(s.image_width)
    bg_c, bsc, bsy, bsx = bg
    view_c, proj_c, campos_c = consts
    has_sh, has_sh_p, has_scales, has_cov = sh is not None, sh_p is not None, scales is not None, cov3D is not None
    M = sh.size(1) if has_sh else 0
    M_p = sh_p.size(1) if has_sh_p else 0
    f32 = dict(device=dev, dtype=torch.float32)
    # (shapes, padded sizes and split of the gradient buffer: the same for every call of a training loop -- computed once)
    plan_key = (P, M, M_p, want_colors, want_cov, has_sh, has_sh_p, has_scales)
    plan = _bw_plans.get(plan_key)
    if plan is None:
        shapes = dict(means3D=(P, 3), means2D=(P, 3), opacities=(P, 1), colors=(P, 3) if want_colors else None,
                      cov3D=(P, 6) if want_cov else None, sh=(P, M, 3) if has_sh else None,
                      sh_p=(P, M_p, 2) if has_sh_p else None, scales=(P, 3) if has_scales else None,
                      rotations=(P, 4) if has_scales else None)
        sizes = {k: (_prod(v) + 3) // 4 * 4 for k, v in shapes.items() if v is not None}
        keys = [k for k, v in shapes.items() if v is not None]
        if len(_bw_plans) > 64:
            _bw_plans.clear()
        plan = _bw_plans[plan_key] = (shapes, sizes, sum(sizes.values()) + 4, keys, [sizes[k] for k in keys] + [4],
                                      tuple(sorted(sizes.items())), [_prod(shapes[k]) for k in keys])
    shapes, sizes, total, keys, split_sizes, layout_key, numels = plan
    zero_buf = None
    entry, reused_grads = None, False
    pool_entry = None              # the kept set of gradient tensors this backward writes (or, second view of a pair, adds to)
    if share_grads is not None:
        pool_entry = share_grads.get("pool_entry")
        # second view of a pair (gftorf_amd.pair): its backward adds to the first view's gradient tensors
        # (cfg.grads_accumulate); only the two scalar offset gradients are its own
        g = {k: v for k, v in share_grads["grads"].items() if k != "offsets"}
    else:
        # One allocation for all per-Gaussian gradients (every tensor a contiguous slice, 16-byte aligned; the two scalar
        # offset gradients at its end): ten torch.empty calls less per forward, which at the reference's scene size
        # (100 k Gaussians, 0.1 ms of kernels per call) is host time the device waits for.  With the zero-fill switch
        # the forward clears it beside its binning kernels and the backward writes only the rows of blended Gaussians.
        entry = None
        if _GRADS_REUSE and pooled and P and pixels is not None and want_bw_records and not zero_fill:
            key = (dev.index, P, layout_key)
            pool = _grad_pool.setdefault(key, [])
            if _USE_COUNT_API:
                # (a buffer somebody wrote to through a tensor -- autograd's in-place sum of two calls' gradients, clipping
                # -- is never trusted again: forgotten as soon as nobody references it)
                pool[:] = [e for e in pool if e["buf"]._version == e["version"] or _storage_refs(e["buf"]) != e["base"]]
                free = [e for e in pool if _storage_refs(e["buf"]) == e["base"] and e["buf"]._version == e["version"]]
            else:
                # (DLPack route: the references that kept autograd from taking the tensors over are dropped now; an alias
                # nobody else holds dies right here and its deleter marks the entry free.  An entry that is still waiting for
                # its backward -- forward A, forward B, A.backward(), B.backward() -- loses that protection before autograd
                # has seen its tensors: they may become a leaf's `.grad` and take the other call's gradient in place, which
                # an alias's dead version counter cannot show.  Such an entry is never trusted to be zero outside its marked
                # rows again: `spoiled` keeps `valid` off, so its next user writes it in full.)
                for e in pool:
                    if e["held"] is not None and not e["valid"]:
                        e["spoiled"] = True
                    e["held"] = None
                free = [e for e in pool if e["free"]]
            # `valid`: the last backward into this buffer returned without error, so every row is defined -- zero or marked
            # in `dirty`.  A buffer that was handed to a forward whose backward never ran (a render under grad used only
            # for logging, a loss skipped by a NaN guard) or failed holds rows nobody wrote: it may be taken again, but as
            # a fresh one that the backward writes in full.
            for e in free:
                if e["valid"]:
                    entry, reused_grads = e, True
                    break
            if entry is None and free:
                entry = free[0]
            if entry is None:
                buf = torch.empty((total,), **f32)
                # (`dirty`: a mark per Gaussian + the 144 bytes behind them in which the rows backward counts; `report`: pinned
                # host word into which it stores the number of rows it wrote -- gft_backward_io.rows_report)
                report = _report_slot()
                entry = dict(buf=buf, dirty=torch.zeros(((P + 3) // 4 * 4 + 144,), device=dev, dtype=torch.uint8), version=buf._version,
                             valid=False, report=report, report_np=report.numpy(), dense_left=0)
                if _USE_COUNT_API:
                    entry["base"] = _storage_refs(buf)
                else:
                    entry.update(mem=buf, buf=None, free=True, held=None)
                pool.append(entry)
                del pool[:-_GRAD_POOL_DEPTH]
                if len(_grad_pool) > 8:
                    _grad_pool.pop(next(iter(_grad_pool)))
            entry["valid"] = False        # until the backward of this forward has returned (run_backward)
            entry["spoiled"] = False
            buf = entry["buf"] if _USE_COUNT_API else _dl_alias(entry)
            pool_entry = entry
        else:
            buf = torch.empty((total,), **f32)
        if zero_fill:
            zero_buf = buf
        # (one split into the padded pieces, then one view each: host time)
        pieces = buf.split(split_sizes)
        g = dict.fromkeys(shapes)
        for k, piece, n in zip(keys, pieces, numels):
            g[k] = (piece if n == sizes[k] else piece[:n]).view(shapes[k])
        g["offsets"] = pieces[-1][:2]
        if entry is not None and not _USE_COUNT_API:
            entry["held"] = (buf, [t for t in g.values() if t is not None])
        if reused_grads and _GRADS_CHECK:
            clean = entry["dirty"][:P] == 0
            for k, t in g.items():
                if t is not None and k != "offsets" and bool(t.reshape(P, -1)[clean].ne(0).any()):
                    raise RuntimeError("gftorf_amd: the kept gradient tensor '%s' holds non-zero rows the last backward did not "
                                       "write -- somebody wrote to it past the version counter (`.data`, a raw pointer); "
                                       "set GFT_GRADS_REUSE=0 for such a caller" % k)
    if "offsets" not in g:
        g["offsets"] = torch.empty((2,), **f32)
    acc_zeroed = acc is not None
    if acc is None:            # second backward through the same forward (retain_graph), or the pybind-level route
        acc = torch.empty((lib.gft_acc_bytes(P) // 4,), **f32)
    cfg = _make_config(s, P, M, M_p, H, W, ph_off, dc_off, (bsc, bsy, bsx), want_bw_records)
    # (2: the backward leaves the accumulator zero again -- the buffer goes back to the pool, _AccLease)
    cfg.acc_zeroed = 2 if (acc_zeroed and acc_lease is not None and _ACC_REUSE and acc_lease.pooled) else int(acc_zeroed)
    # A kept set of gradient tensors is rewritten row by row only while few rows are written (a dense frame blends a few
    # per cent of its Gaussians): the rows kernel stores its rows straight from the lanes, and from about a third of the
    # Gaussians on the full write through LDS -- coalesced, zeros included -- is faster (C3-shaped frame, 96 % blended: 16 vs
    # 54 us; fog: 140 vs 203 us).  The last rows backward has left its row count in pinned memory: above _DENSE_SHARE the
    # next _DENSE_RUN backwards into these tensors write them in full (which marks every row), then one rows backward
    # looks again.
    rows_only = reused_grads
    sample = False                 # this backward reports its row count (the report costs the rows kernel ~4 us: every 16th call)
    if reused_grads and pool_entry is not None and "report_np" in pool_entry:
        e = pool_entry
        if e["dense_left"] > 0:
            e["dense_left"] -= 1
            rows_only = False
            e["probe"] = e["dense_left"] == 0
        elif int(e["report_np"][0]) > _DENSE_SHARE * P:
            e["dense_left"] = _DENSE_RUN
            e["report_np"][0] = 0
            rows_only = False
        else:
            e["tick"] = e.get("tick", 0) + 1
            sample = e.get("probe", False) or e["tick"] <= 2 or e["tick"] % 16 == 0
            e["probe"] = False
    cfg.grads_zeroed = 3 if rows_only else int(zero_buf is not None)
    cfg.grads_accumulate = int(share_grads is not None)
    if share_grads is not None and share_grads.get("dirty") is not None:
        # second view of a pair: its rows are added to the first view's tensors and marked in the same array
        entry = dict(dirty=share_grads["dirty"])
    io = _lib.BackwardIO()
    # the two scalar gradients cost a reduction launch: only when the caller optimises an offset (the reference returns
    # None for them otherwise, __init__.py:202-203); the pybind-level route always returns them
    want_off = getattr(s, "optimize_phase_offset", True) or getattr(s, "optimize_dc_offset", True)
    off_ptr = g["offsets"].data_ptr() if want_off else 0
    # (cfg.grads_zeroed = 3: the backward zeroes the rows the previous one wrote and this one does not, then writes its own)
    dirty_ptr = entry["dirty"].data_ptr() if entry is not None else 0
    report_ptr = entry["report"].data_ptr() if (entry is not None and sample and entry.get("report") is not None) else 0
    # (the argument block in one pack, field order of gft_backward_io; the upstream gradients, binning and det_partials follow
    # in run_backward)
    _struct.pack_into(_BWD_FMT, io, 0,
                      _p0(bg_c), means3D.data_ptr() if P else 0, radii.data_ptr() if P else 0,
                      _p0(scales) if has_scales else 0, _p0(rotations) if has_scales else 0, _p0(cov3D) if has_cov else 0,
                      view_c.data_ptr(), proj_c.data_ptr(), campos_c.data_ptr(), _p0(sh) if has_sh else 0, _p0(sh_p) if has_sh_p else 0,
                      _p0(opac) if P else 0, _p0(pixels) if P else 0,
                      0, 0, 0, 0, 0,
                      geom.data_ptr(), img.data_ptr(), 0, acc.data_ptr() if P else 0,
                      g["means3D"].data_ptr() if P else 0, g["means2D"].data_ptr() if P else 0, _p0(g["colors"]),
                      g["opacities"].data_ptr() if P else 0, _p0(g["cov3D"]), _p0(g["sh"]), _p0(g["sh_p"]), _p0(g["scales"]),
                      _p0(g["rotations"]), off_ptr, off_ptr + 4 if off_ptr else 0, 0, dirty_ptr, report_ptr)
    last_call_stats["grads_reused"] = bool(reused_grads)
    last_call_stats["grads_rows_only"] = bool(rows_only)
    return dict(grads=g, cfg=cfg, io=io, acc=acc, acc_lease=acc_lease, pixels=pixels, zero_buf=zero_buf, dev=dev, P=P, H=H, W=W,
                dirty=entry["dirty"] if entry is not None else None, pool_entry=pool_entry,
                debug_args=(s.bg, means3D, radii, scales, rotations, s.scale_modifier, cov3D, s.viewmatrix, s.projmatrix,
                            s.tanfovx, s.tanfovy, sh, sh_p, s.sh_degree, s.campos, s.debug, s.near_n, s.far_n, s.depth_range,
                            s.use_view_dependent_phase, ph_off, dc_off) if s.debug else None)


def run_backward(prep, grads_out, geom, binning, img, debug=False):
    """The rest of a backward: the upstream gradients (color, phasor, depth, acc, depth_distortion; None = zeros) and the
    launch (``RasterizeGaussiansBackwardCUDA``, rasterize_points.cu:167-281).  Returns the dict of gradient tensors."""
    lib = _lib.load()
    dev, P, H, W, io = prep["dev"], prep["P"], prep["H"], prep["W"], prep["io"]

    def gr(t, c, name):
        # gradients of normal / entropy / amp_distortion / pixels / distribution are
        # accepted and ignored, as in the reference kernels (backward.cu:609-630)
        if t is None:
            return None
        if t.dim() != 3 or t.size(0) != c or t.size(1) != H or t.size(2) != W:
            raise RuntimeError("grad of %s has shape %s, expected %s" % (name, tuple(t.shape), (c, H, W)))
        return _f32(t, dev, "grad_" + name)

    keep = (gr(grads_out[0], 3, "color"), gr(grads_out[1], 7, "phasor"), gr(grads_out[2], 1, "depth"),
            gr(grads_out[3], 1, "acc"), gr(grads_out[4], 1, "depth_distortion"))
    io.dL_dout_color, io.dL_dout_phasor = _ptr(keep[0]), _ptr(keep[1])
    io.dL_dout_depth, io.dL_dout_acc, io.dL_dout_depth_distortion = _ptr(keep[2]), _ptr(keep[3]), _ptr(keep[4])
    io.geom, io.img, io.binning = _ptr(geom), _ptr(img), _ptr(binning)
    det = None
    cap = prep.get("cap")              # (known to the forward that made `prep`; else read back from the buffer's size)
    if cap is None:
        cap = binning_capacity(binning, W, H) if P else 0
    if _DETERMINISTIC and P and cap:
        # test mode: partial rows per (list entry, quadrant), added in a fixed order (gft_backward_io.det_partials)
        # (capturable since round 6: what gave wrong sums from the second replay on was the library's hipMemsetAsync of this
        # buffer -- a large memset node is not ordered against its neighbours when a graph is replayed on this platform; the
        # library clears with a kernel now, gft_api.hip gft_zero_async)
        det = torch.empty((lib.gft_det_partials_bytes(cap, W, H) // 4,), dtype=torch.float32, device=dev)
        io.det_partials = det.data_ptr()
    if debug:
        cpu_args = cpu_deep_copy_tuple(prep["debug_args"] + tuple(grads_out) + (geom, binning, img))
    lease = prep.get("acc_lease")
    if lease is not None:
        lease.zero = False             # the render backward writes to it; zero again only if the call below returns
    pool_entry = prep.get("pool_entry")
    try:
        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            _lib.check(lib.gft_backward(stream, C.byref(prep["cfg"]), C.byref(io), cap))
        if pool_entry is not None and not prep["cfg"].grads_accumulate:
            # every row of the kept gradient tensors is defined now (api._grad_pool) -- unless, on the DLPack route, another
            # forward of the shape let go of them before this backward ran
            pool_entry["valid"] = not pool_entry.get("spoiled", False)
        if lease is not None and prep["cfg"].acc_zeroed == 2:
            lease.zero, lease.stream = True, stream
            lease.give_back()
    except Exception as ex:
        if pool_entry is not None:
            pool_entry["valid"] = False
        if debug:
            torch.save(cpu_args, "snapshot_bw.dump")
            print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        raise ex
    return prep["grads"]


def native_backward(s, means3D, opac, sh, sh_p, scales, rotations, cov3D, radii, geom, binning, img, bg, consts,
                    ph_off, dc_off, grads_out, acc, want_colors, want_cov, want_bw_records=True):
    """One backward of the native rasterizer: :func:`prepare_backward` + :func:`run_backward` in one go."""
    prep = prepare_backward(s, means3D, opac, sh, sh_p, scales, rotations, cov3D, radii, geom, img, bg, consts, ph_off,
                            dc_off, acc, want_colors, want_cov, want_bw_records)
    return run_backward(prep, grads_out, geom, binning, img, bool(s.debug))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_p, colors_precomp, phasors_precomp, opacities,
                scales, rotations, cov3Ds_precomp, phase_offset, dc_offset, raster_settings):
        s = raster_settings
        ph_off = _scalar(phase_offset)
        dc_off = _scalar(dc_offset)
        # the backward can only run if autograd tracks one of the inputs
        want_bw = any(ctx.needs_input_grad)
        r = native_forward(s, means3D, sh, sh_p, colors_precomp, phasors_precomp, opacities, scales, rotations,
                           cov3Ds_precomp, ph_off, dc_off, want_bw, True)
        means3D_c, opac_c, sh_c, sh_p_c, scales_c, rot_c, cov_c, colors_c, phasors_c = r["inputs"]
        ctx.raster_settings = s
        ctx.num_rendered = r["R"]
        ctx.binning_instances = r["cap"]
        ctx.scalars = (ph_off, dc_off)
        ctx.want_bw = want_bw
        ctx.acc = r["acc"]         # zeroed, valid for the first backward of this forward
        ctx.prep = r["prep"]       # gradient tensors + argument block of that first backward
        ctx.bg = r["bg"]
        ctx.consts = r["consts"]
        ctx.present = (sh_c is not None, sh_p_c is not None, colors_c is not None, phasors_c is not None,
                       scales_c is not None, cov_c is not None)
        ctx.in_shapes = (opacities.shape, phase_offset.shape if isinstance(phase_offset, torch.Tensor) else None,
                         dc_offset.shape if isinstance(dc_offset, torch.Tensor) else None)
        ctx.set_materialize_grads(False)
        dummy = means3D_c.new_empty(0)
        radii = r["outputs"][10]
        ctx.save_for_backward(means3D_c, opac_c if opac_c is not None else dummy,
                              sh_c if sh_c is not None else dummy, sh_p_c if sh_p_c is not None else dummy,
                              scales_c if scales_c is not None else dummy, rot_c if rot_c is not None else dummy,
                              cov_c if cov_c is not None else dummy, radii, r["geom"], r["binning"], r["img"],
                              r["outputs"][8])     # `pixels`: the backward reads it (which Gaussians were blended), so an
                                                   # in-place edit between forward and backward must raise, not zero rows
        ctx.mark_non_differentiable(radii)
        return r["outputs"]

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_phasor, grad_out_depth, grad_out_normal, grad_out_acc,
                 grad_entropy, grad_depth_distortion, grad_amp_distortion, grad_pixels, grad_distribution, _):
        s = ctx.raster_settings
        means3D, opac, sh, sh_p, scales, rotations, cov3D, radii, geom, binning, img, _pixels = ctx.saved_tensors
        has_sh, has_sh_p, has_colors, has_phasors, has_scales, has_cov = ctx.present
        ph_off, dc_off = ctx.scalars
        acc, ctx.acc = ctx.acc, None
        prep, ctx.prep = ctx.prep, None
        grads_out = (grad_out_color, grad_out_phasor, grad_out_depth, grad_out_acc, grad_depth_distortion)
        if prep is not None:
            g = run_backward(prep, grads_out, geom, binning, img, bool(s.debug))
        else:      # a second backward through the same forward (retain_graph): fresh tensors, own accumulator clear
            g = native_backward(s, means3D, opac, sh if has_sh else None, sh_p if has_sh_p else None,
                                scales if has_scales else None, rotations if has_scales else None,
                                cov3D if has_cov else None, radii, geom, binning, img, ctx.bg, ctx.consts, ph_off, dc_off,
                                grads_out, None, has_colors, has_cov, ctx.want_bw)
        op_shape, ph_shape, dc_shape = ctx.in_shapes
        grad_phase = grad_dc = None
        if s.optimize_phase_offset and ph_shape is not None:
            grad_phase = g["offsets"][0:1].reshape(ph_shape)
        if s.optimize_dc_offset and dc_shape is not None:
            grad_dc = g["offsets"][1:2].reshape(dc_shape)
        # input order of forward(); the reference has no backward for phasors_precomp
        return (g["means3D"], g["means2D"], g["sh"], g["sh_p"], g["colors"], None,
                g["opacities"].reshape(op_shape), g["scales"], g["rotations"], g["cov3D"],
                grad_phase, grad_dc, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: near_n <= view-space z <= far_n (reference rasterizer_impl.cu:54-68)."""
        with torch.no_grad():
            s = self.raster_settings
            lib = _lib.load()
            if positions.device.type != "cuda":
                raise RuntimeError("gftorf_amd: markVisible needs HIP tensors; there is no CPU path")
            dev = positions.device
            pos = _f32(positions, dev, "positions")
            P = pos.size(0)
            visible = torch.zeros((P,), device=dev, dtype=torch.bool)
            if P:
                view = _f32(s.viewmatrix, dev, "viewmatrix")
                proj = _f32(s.projmatrix, dev, "projmatrix")
                with _lib.on_device(dev):
                    _lib.check(lib.gft_mark_visible(_lib.raw_stream(dev), P, pos.data_ptr(),
                                                    view.data_ptr(), proj.data_ptr(), float(s.near_n),
                                                    float(s.far_n), visible.data_ptr()))
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, shs_p=None, colors_precomp=None,
                phasors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                phase_offset=0.0, dc_offset=0.0):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        # (the reference builds a fresh `torch.Tensor([])` for every absent input on every call; one shared empty tensor
        # says the same -- it is never written to -- and saves the host ~2 us each)
        if shs is None:
            shs = _EMPTY
        if colors_precomp is None:
            colors_precomp = _EMPTY
        if shs_p is None:
            shs_p = _EMPTY
        if phasors_precomp is None:
            phasors_precomp = _EMPTY
        if scales is None:
            scales = _EMPTY
        if rotations is None:
            rotations = _EMPTY
        if cov3D_precomp is None:
            cov3D_precomp = _EMPTY

        return rasterize_gaussians(means3D, means2D, shs, shs_p, colors_precomp, phasors_precomp, opacities,
                                   scales, rotations, cov3D_precomp, phase_offset, dc_offset, raster_settings)
