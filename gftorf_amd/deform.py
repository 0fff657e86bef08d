"""The deformation network on the gfx950 matrix cores.

Host-side mirror of the reference's ``DeformNetwork`` (``utils/time_utils.py:56-127``): same
constructor, same parameter names and shapes (a ``deform_model.pth`` written by the reference's
``DeformModel.save`` loads with ``load_state_dict``, ``scene/deform_model.py:35-40``), same four
results of ``forward(x, t)``:

    d_xyz [n,3], zeros [n,4], d_sh [n,16,3], zeros [n,16,2]

(the reference returns zeros for the rotation and phasor offsets, ``time_utils.py:127``, and its
``rot`` / ``a`` heads never reach an output: their parameters exist here for the state_dict and,
as under the reference's autograd, never receive a gradient).

The arithmetic is done by hand-written MFMA kernels (``csrc/k_deform.hip``) behind the C ABI of
``include/gftorf_deform.h`` -- fp32 results from two fp16 (or three bf16) planes per operand, or fp32-operand MFMA with
``GFT_DEFORM_BF16X3=0``; there is no CPU path and no torch GEMM in it.  Inputs are not
differentiated: the reference passes detached positions (``scene/gaussian_model.py:172``).
"""
import ctypes as C

import torch
from torch import nn

from . import _lib

_D, _W, _SH_DEGREE = 8, 256, 3
_PAD = 192                       # DF_PAD: point counts are padded to multiples of 192 inside the saved / scratch buffers

# The backward runs over the points whose upstream gradient row (d_xyz, d_sh) is non-zero only: in a training iteration
# those are the Gaussians some pixel blended (6-14 % of the queried points on the metric scene); a row whose upstream
# gradient is zero has dz = 0 in every layer and adds exactly nothing to any weight gradient.  The saved activations of
# the rows that count are compacted (one rank computation, one blocking read of the count -- as the reference's
# `t[mask]` has), then the same kernels run on the compact set: gradients equal the dense backward's up to summation
# order.  Off: `gftorf_amd.deform.sparse_backward = False`.
sparse_backward = True
_SPARSE_MIN_POINTS = 8192        # below this the dense backward is launch-bound anyway
_SPARSE_MAX_FRACTION = 0.6       # above this the compaction costs more than it saves
last_backward_stats = {"points": 0, "points_processed": 0, "recomputed": False}


def backward_stats():
    """``last_backward_stats`` with the row count as a number: a backward that counted its rows on the device
    (``device_row_count``) leaves the count there (``rows_on_device``); this reads it (one blocking read)."""
    st = dict(last_backward_stats)
    rows = st.pop("rows_on_device", None)
    if st.get("points_processed") is None:
        st["points_processed"] = int(rows.item()) if rows is not None else 0
    return st

# Activations on demand.  A saving forward writes 8.4 KB per point for the backward (2.5 GB at 300 k points, a sixth of the
# forward's time -- and of its energy: that kernel runs at the board's power limit).  When the last backward used few of
# the rows (the usual training iteration: only Gaussians that a pixel blended carry a gradient), the next forward keeps
# NOTHING; its backward gathers the inputs of the rows that count, runs the saving forward on those rows alone (a point's
# activations do not depend on the batch it is in: bit-identical) and goes on as before -- no 2.5 GB, no compaction of them.
# (Bit-identical while the call stays inside the fp16 planes' range, which is every network the reference trains: the
# library decides per CALL whether the fp16 walk's results stand or the fp32-range walk overwrites them (k_deform.hip,
# range guard).  A forward that fell back because of a row WITHOUT a gradient is recomputed on the rows with one by the fp16
# walk: the two walks agree to ~1e-7, a ReLU input within that of zero may change side.)
# A backward that finds many rows with a gradient after all recomputes all of them (one extra forward, once) and the next
# forward saves again.  `GFT_DEFORM_LAZY_SAVE=0` in the environment or `gftorf_amd.deform.lazy_save = False`: always save.
import os as _os
lazy_save = _os.environ.get("GFT_DEFORM_LAZY_SAVE", "1") != "0"
_LAZY_MAX_FRACTION = 0.25        # recomputing that share of the rows costs less than saving all of them

# Rows counted on the device.  The row selection above reads a count back (one blocking read, as the reference's `t[mask]`
# has), which a stream that is being captured into a HIP graph cannot do.  ``gft_deform_backward_rows`` keeps the count on the
# device: the forward keeps nothing, the backward marks, ranks and counts the rows with a gradient by kernels, gathers their
# inputs, recomputes their activations and runs over them -- launches of the capacity whose surplus workgroups return at
# once.  A captured iteration replayed on other gradients adapts to THEIR rows (the C3 loop from a graph: the backward
# follows the share of Gaussians a pixel blends, 58 % -> 23 % over the run, instead of staying dense).  Gradients are those of
# the blocking selection bit for bit (tests/test_deform.py).
#   "auto" (default): under capture always; eagerly whenever the work buffer (sized for ALL queried points: 17 KB each) stays
#       below ``_DEVICE_ROWS_MAX_BYTES`` -- the share of rows with a gradient is then learnt WITHOUT a blocking read: every
#       backward leaves its count in pinned memory and the next forward looks at whatever has arrived.  While that share is
#       unknown or above ``_DEVICE_ROWS_MAX_FRACTION`` the forward saves its activations and the backward runs dense (and
#       counts); below it the forward keeps nothing and the backward recomputes the rows that count.  No host read anywhere.
#   "capture": under capture only (eagerly the blocking selection above);  True: always;  False: never (a captured call
#       then saves its activations and runs dense).
# ``GFT_DEFORM_DEVICE_ROWS`` = auto | capture | 1 | 0.
device_row_count = {"0": False, "1": True, "capture": "capture", "auto": "auto"}.get(_os.environ.get("GFT_DEFORM_DEVICE_ROWS", ""), "auto")
_DEVICE_ROWS_MAX_FRACTION = 0.6
_DEVICE_ROWS_MAX_BYTES = 32 << 30


def _peek_fraction(state):
    """The share of rows the last counted backward used, if its count has reached the host by now (never waits)."""
    pend = state.get("pending") if state is not None else None
    if pend is not None and pend[1].query():
        state["fraction"] = int(pend[0].item()) / float(max(pend[2], 1))
        state["pending"] = None


def _post_count(state, rows, n_all):
    """Queues the device count's way to the host (pinned memory, no wait): the next forward's `_peek_fraction` reads it."""
    if state is None or torch.cuda.is_current_stream_capturing():
        return
    pin = state.get("pin")
    if pin is None:
        pin = state["pin"] = torch.zeros((1,), dtype=torch.int32).pin_memory()
    pin.copy_(rows, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    state["pending"] = (pin, ev, n_all)


def _param_list(mod):
    ps = []
    for l in mod.linear:
        ps += [l.weight, l.bias]
    for h in (mod.xyz_warp, mod.r, mod.g, mod.b):
        ps += [h.weight, h.bias]
    return ps


def _fill(struct, tensors):
    """24 tensors in _param_list order -> gft_deform_params / gft_deform_grads."""
    for i in range(_D):
        struct.linear_w[i] = tensors[2 * i].data_ptr()
        struct.linear_b[i] = tensors[2 * i + 1].data_ptr()
    k = 2 * _D
    (struct.xyz_w, struct.xyz_b, struct.r_w, struct.r_b, struct.g_w, struct.g_b, struct.b_w, struct.b_b) = [
        t.data_ptr() for t in tensors[k:k + 8]]
    return struct


def _prod(shape):
    k = 1
    for d in shape:
        k *= int(d)
    return k


class _DeformFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xm, tm, state, x, t, *params):
        lib = _lib.load()
        dev = x.device
        if dev.type != "cuda":
            raise RuntimeError("gftorf_amd.DeformNetwork runs on a HIP device only (x is on %s); there is no CPU path" % (dev,))
        if x.dim() != 2 or x.size(1) != 3:
            raise RuntimeError("x must have dimensions (num_points, 3)")
        n = x.size(0)
        if t.numel() != n and t.numel() != 1:
            raise RuntimeError("t must hold one value per point, got %s for %d points" % (tuple(t.shape), n))
        x_c = x.detach().float().contiguous()
        if t.numel() == 1 or (t.dim() >= 1 and t.stride(0) == 0):
            t_c, t_stride = t.detach().float().reshape(-1)[:1].contiguous(), 0       # one time for all points
        else:
            t_c, t_stride = t.detach().float().reshape(-1).contiguous(), 1
        if t_c.device != dev:
            raise RuntimeError("t is on %s, expected %s" % (t_c.device, dev))
        ps = []
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError("DeformNetwork parameters must be float32 on %s" % (dev,))
            ps.append(p.detach().contiguous())
        need_bw = any(ctx.needs_input_grad[5:])      # all False under torch.no_grad()
        # (while the stream is being captured into a graph nothing may be read from the device: the blocking row selection
        # reads a row count, so a captured call either counts its rows on the device -- below -- or, with
        # device_row_count = False, saves its activations and runs the dense backward: the same gradients up to summation order)
        capturing = torch.cuda.is_current_stream_capturing()
        sparse_ok = bool(need_bw and sparse_backward and n >= _SPARSE_MIN_POINTS)
        # rows counted on the device (see device_row_count): "rows" = the forward keeps nothing, the backward recomputes the rows
        # that count; "count" = the forward saves, the backward runs dense and only counts (the share is unknown or large)
        dev_mode = None
        if sparse_ok and device_row_count is not False:
            if device_row_count is True or (capturing and device_row_count in ("capture", "auto")):
                dev_mode = "rows"
            elif device_row_count == "auto" and state is not None and lib.gft_deform_rows_work_bytes(n) <= _DEVICE_ROWS_MAX_BYTES:
                _peek_fraction(state)
                frac = state.get("fraction")
                dev_mode = "rows" if (frac is not None and frac <= _DEVICE_ROWS_MAX_FRACTION and lazy_save) else "count"
        dev_rows = dev_mode == "rows"
        lazy = bool(sparse_ok and lazy_save and state is not None and dev_mode is None
                    and state.get("fraction") is not None and state["fraction"] <= _LAZY_MAX_FRACTION and not capturing) or dev_rows
        f32 = dict(device=dev, dtype=torch.float32)
        packed = torch.empty((lib.gft_deform_packed_bytes() // 4,), **f32)
        d_xyz = torch.empty((n, 3), **f32)
        d_sh = torch.empty((n, 16, 3), **f32)
        saved = torch.empty((lib.gft_deform_saved_bytes(n) // 4,), **f32) if (need_bw and n > 0 and not lazy) else None
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            _lib.check(lib.gft_deform_pack(stream, xm, tm, C.byref(_fill(_lib.DeformParams(), ps)), packed.data_ptr()))
            _lib.check(lib.gft_deform_forward(stream, xm, tm, n, x_c.data_ptr() if n else None, t_c.data_ptr() if n else None,
                                              t_stride, packed.data_ptr(), saved.data_ptr() if saved is not None else None,
                                              d_xyz.data_ptr() if n else None, d_sh.data_ptr() if n else None))
        ctx.n = n
        ctx.arch = (xm, tm)
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.state = state
        ctx.lazy = lazy
        ctx.dev_rows = dev_rows
        ctx.dev_count = dev_mode == "count"
        if lazy:
            ctx.t_stride = t_stride
            ctx.save_for_backward(packed, x_c, t_c)
        else:
            ctx.save_for_backward(packed, saved if saved is not None else packed.new_empty(0))
        ctx.set_materialize_grads(False)
        return d_xyz, d_sh

    @staticmethod
    def backward(ctx, g_dxyz, g_dsh):
        lib = _lib.load()
        if ctx.lazy:
            packed, x_c, t_c = ctx.saved_tensors
            saved = None
        else:
            packed, saved = ctx.saved_tensors
        dev = packed.device
        n = ctx.n
        if n > 0 and not ctx.lazy and saved.numel() == 0:
            raise RuntimeError("DeformNetwork: backward through a forward that ran without gradients")
        f32 = dict(device=dev, dtype=torch.float32)
        # the gradients are views of ONE flat buffer, in parameter order: the data-parallel bucket (flat_grad_bucket) is
        # then this memory itself and the all-reduce needs no gather / scatter copies
        # (each view starts on a 16-byte boundary, as a tensor of its own would; the few padding floats stay zero)
        sizes = [_prod(s) for s in ctx.shapes]
        flat = torch.zeros((sum((k + 3) // 4 * 4 for k in sizes),), **f32)
        grads, o = [], 0
        for shp, k in zip(ctx.shapes, sizes):
            grads.append(flat[o:o + k].view(shp))
            o += (k + 3) // 4 * 4
        gx = g_dxyz.float().contiguous() if g_dxyz is not None else None
        gs = g_dsh.float().contiguous() if g_dsh is not None else None
        last_backward_stats.update(points=n, points_processed=n, recomputed=ctx.lazy)
        last_backward_stats.pop("rows_on_device", None)       # (a count an earlier backward left on the device)
        n_all = n
        if ctx.dev_rows:
            # rows counted on the device: one call, nothing read back (capturable)
            work = torch.empty((lib.gft_deform_rows_work_bytes(n) // 4,), **f32)
            rows = torch.empty((1,), device=dev, dtype=torch.int32)
            ptr = lambda t_: t_.data_ptr() if t_ is not None else None
            with _lib.on_device(dev):
                _lib.check(lib.gft_deform_backward_rows(_lib.raw_stream(dev), ctx.arch[0], ctx.arch[1], n, packed.data_ptr(), x_c.data_ptr(),
                                                        t_c.data_ptr(), ctx.t_stride, ptr(gx), ptr(gs), work.data_ptr(),
                                                        C.byref(_fill(_lib.DeformParams(), grads)), rows.data_ptr()))
            last_backward_stats.update(points_processed=None, rows_on_device=rows)
            _post_count(ctx.state, rows, n_all)
            return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:]))
        if ctx.dev_count and (gx is not None or gs is not None):
            # dense backward over the saved activations; the rows with a gradient are only counted (three small launches, the
            # count on its way to pinned memory): the next forward decides by it
            rows = _count_rows_on_device(lib, n, gx, gs, dev)
            _post_count(ctx.state, rows, n_all)
            last_backward_stats.pop("rows_on_device", None)
        if ctx.lazy:
            n, saved, gx, gs = _recompute_rows(lib, ctx, n, packed, x_c, t_c, gx, gs)
            last_backward_stats["points_processed"] = n
        elif (sparse_backward and n >= _SPARSE_MIN_POINTS and (gx is not None or gs is not None) and not ctx.dev_count
              and not torch.cuda.is_current_stream_capturing()):
            n, saved, gx, gs = _compact_rows(lib, n, saved, gx, gs)
            last_backward_stats["points_processed"] = n
        if ctx.state is not None and n_all >= _SPARSE_MIN_POINTS and not ctx.dev_count and not torch.cuda.is_current_stream_capturing():
            ctx.state["fraction"] = n / float(n_all)
        scratch = torch.empty((lib.gft_deform_scratch_bytes(n) // 4,), **f32)
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            _lib.check(lib.gft_deform_backward(stream, ctx.arch[0], ctx.arch[1], n, packed.data_ptr(), saved.data_ptr() if n else None,
                                               gx.data_ptr() if (gx is not None and n) else None,
                                               gs.data_ptr() if (gs is not None and n) else None,
                                               scratch.data_ptr() if n else None,
                                               C.byref(_fill(_lib.DeformParams(), grads))))
        return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:]))


def _count_rows_on_device(lib, n, gx, gs, dev):
    """int32 [1] on the device: the number of rows whose upstream gradient holds a value != 0; nothing read back."""
    mask = torch.empty((n,), device=dev, dtype=torch.uint8)
    rank = torch.empty((n,), device=dev, dtype=torch.int32)
    scratch = torch.empty((lib.gft_rows_rank_scratch_bytes(n),), device=dev, dtype=torch.uint8)
    rows = torch.empty((1,), device=dev, dtype=torch.int32)
    ptr0 = lambda t: t.data_ptr() if t is not None else None
    with _lib.on_device(dev):
        stream = _lib.raw_stream(dev)
        _lib.check(lib.gft_rows_any_nonzero(stream, n, 3 if gx is not None else 0, ptr0(gx), 48 if gs is not None else 0, ptr0(gs),
                                            mask.data_ptr()))
        _lib.check(lib.gft_rows_rank_dev(stream, n, mask.data_ptr(), rank.data_ptr(), scratch.data_ptr(), rows.data_ptr()))
    return rows


def _rows_with_gradient(lib, n, gx, gs, dev):
    """RowSelection of the rows whose upstream gradient holds a value != 0 (a NaN row counts, so a diverged step shows in
    the weight gradients as it does with the dense backward and in the reference): one pass over the two tensors."""
    from .densify import RowSelection
    mask = torch.empty((n,), device=dev, dtype=torch.uint8)
    ptr0 = lambda t: t.data_ptr() if t is not None else None
    with _lib.on_device(dev):
        _lib.check(lib.gft_rows_any_nonzero(_lib.raw_stream(dev), n, 3 if gx is not None else 0, ptr0(gx),
                                            48 if gs is not None else 0, ptr0(gs), mask.data_ptr()))
    return RowSelection(mask.view(torch.bool))


def _recompute_rows(lib, ctx, n, packed, x_c, t_c, gx, gs):
    """The forward kept nothing (lazy_save): the saved activations of the rows with an upstream gradient, from a saving
    forward over those rows alone (all rows when they turn out to be many), and the gradients of those rows."""
    dev = packed.device
    f32 = dict(device=dev, dtype=torch.float32)
    if n == 0 or (gx is None and gs is None):
        return 0, None, None, None
    sel = _rows_with_gradient(lib, n, gx, gs, dev)
    k = sel.count
    if k == 0:
        return 0, None, None, None
    if k > _SPARSE_MAX_FRACTION * n:
        k, xs, ts = n, x_c, t_c                                      # dense after all: every row again
    else:
        xs = sel.take(x_c)
        ts = t_c if ctx.t_stride == 0 else sel.take(t_c)
        gx = sel.take(gx) if gx is not None else None
        gs = sel.take(gs) if gs is not None else None
    saved = torch.empty((lib.gft_deform_saved_bytes(k) // 4,), **f32)
    d_xyz, d_sh = torch.empty((k, 3), **f32), torch.empty((k, 16, 3), **f32)
    with _lib.on_device(dev):
        _lib.check(lib.gft_deform_forward(_lib.raw_stream(dev), ctx.arch[0], ctx.arch[1], k, xs.data_ptr(), ts.data_ptr(),
                                          ctx.t_stride, packed.data_ptr(), saved.data_ptr(), d_xyz.data_ptr(), d_sh.data_ptr()))
    return k, saved, gx, gs


def _compact_rows(lib, n, saved, gx, gs):
    """Rows of (saved activations, upstream gradients) whose upstream gradient is non-zero, compacted; everything
    unchanged when most rows count.  `saved` = [n_pad, E] encoding | [8, n_pad, 256] activations | [8, n_pad, 8] ReLU sign
    words (gft_deform_saved_bytes), n_pad = n rounded up to 192."""
    sel = _rows_with_gradient(lib, n, gx, gs, saved.device)
    k = sel.count
    if k > _SPARSE_MAX_FRACTION * n:
        return n, saved, gx, gs
    if k == 0:
        return 0, saved, None, None
    dev = saved.device
    f32 = dict(device=dev, dtype=torch.float32)
    out = torch.empty((lib.gft_deform_saved_bytes(k) // 4,), **f32)
    idx = torch.empty((k,), device=dev, dtype=torch.int32)
    gx_c = torch.empty((k, 3), **f32) if gx is not None else None
    gs_c = torch.empty((k, 16, 3), **f32) if gs is not None else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    with _lib.on_device(dev):
        _lib.check(lib.gft_deform_compact(_lib.raw_stream(dev), n, k, sel.mask.data_ptr(), sel.rank.data_ptr(), saved.data_ptr(),
                                          ptr(gx), ptr(gs), idx.data_ptr(), out.data_ptr(), ptr(gx_c), ptr(gs_c)))
    return k, out, gx_c, gs_c


class DeformNetwork(nn.Module):
    """Drop-in for ``utils.time_utils.DeformNetwork`` (time_utils.py:56-127).

    The reference constructs it as ``DeformNetwork(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3)``
    (``scene/deform_model.py:9-16`` from ``arguments/__init__.py:66-69``; ``configs/torf.json:9-12`` and
    ``configs/ftorf.json`` say the same): 84 encoded inputs, ``linear.0.weight [256, 84]``,
    ``linear.5.weight [256, 340]``, 522 055 parameters.  The signature's own default ``t_multires=6`` (76
    inputs) is kept as the reference has it.  Any octave counts whose encoding has at most 96 columns run on
    the same kernels; D, W and the SH degree are those of every shipped configuration."""

    def __init__(self, D=8, W=256, xyz_multires=10, t_multires=6, sh_degree=3):
        super().__init__()
        n_in = 3 + 6 * xyz_multires + 1 + 2 * t_multires
        if (D, W, sh_degree) != (_D, _W, _SH_DEGREE) or xyz_multires < 0 or t_multires < 0 or n_in > _lib.DEFORM_MAX_INPUTS:
            raise NotImplementedError(
                "gftorf_amd.DeformNetwork is built for D=8, W=256, sh_degree=3 (every configuration the reference "
                "ships) and encodings of at most %d columns (xyz_multires=10 with t_multires=10 gives 84); got %s"
                % (_lib.DEFORM_MAX_INPUTS, (D, W, xyz_multires, t_multires, sh_degree)))
        _IN = n_in
        self.D, self.W = D, W
        self.xyz_multires, self.t_multires = xyz_multires, t_multires
        self.skips = [D // 2]
        self.xyz_input_ch, self.t_input_ch = 3 + 6 * xyz_multires, 1 + 2 * t_multires
        self.num_shs = (1 + sh_degree) ** 2
        # layer i + 1 takes the re-injected encoding when i is a skip (time_utils.py:69-73)
        widths = [_IN] + [W + _IN if i in self.skips else W for i in range(D - 1)]
        self.linear = nn.ModuleList([nn.Linear(w_in, W) for w_in in widths])
        self.xyz_warp = nn.Linear(W, 3)
        self.rot = nn.Linear(W, 4)
        self.r = nn.Linear(W, self.num_shs)
        self.g = nn.Linear(W, self.num_shs)
        self.b = nn.Linear(W, self.num_shs)
        self.a = nn.Linear(W, self.num_shs)
        self.isotropic = False

    def initialize_weights(self, args):
        """time_utils.py:83-101: Xavier-normal trunk with zero biases, heads N(0, 1e-5) (``xyz_warp``
        Xavier-normal instead when ``args.xavier_init_dxyz``)."""
        self.isotropic = args.isotropic_gaussians
        for l in self.linear:
            nn.init.xavier_normal_(l.weight)
            nn.init.constant_(l.bias, 0.0)
        if args.xavier_init_dxyz:
            nn.init.xavier_normal_(self.xyz_warp.weight)
        else:
            nn.init.normal_(self.xyz_warp.weight, mean=0.0, std=1e-5)
        nn.init.constant_(self.xyz_warp.bias, 0.0)
        for head in (self.rot, self.r, self.g, self.b, self.a):
            nn.init.normal_(head.weight, mean=0.0, std=1e-5)
            nn.init.constant_(head.bias, 0.0)

    def forward(self, x, t, zeros_as_scalars=False):
        """``d_xyz, d_rot, d_sh, d_sh_p`` as the reference returns them (time_utils.py:127: the rotation and phasor offsets
        are zeros).  ``zeros_as_scalars=True`` returns the Python float 0.0 for those two instead of [n, 4] and [n, 16, 2]
        tensors of zeros -- what train.py:164 passes for a static scene, and what the renderer's additions and
        ``assemble_inputs`` take as well: no 128 bytes per point filled, read and given a gradient for nothing."""
        if not hasattr(self, "_save_state"):
            self._save_state = {"fraction": None, "pending": None, "pin": None}      # share of the rows the last backward used
        d_xyz, d_sh = _DeformFn.apply(self.xyz_multires, self.t_multires, self._save_state, x, t, *_param_list(self))
        if zeros_as_scalars:
            return d_xyz, 0.0, d_sh, 0.0
        n = x.size(0)
        zeros = lambda *shape: torch.zeros(shape, device=x.device, dtype=torch.float32)
        return d_xyz, zeros(n, 4), d_sh, zeros(n, self.num_shs, 2)


# scene/deform_model.py:9-16 with the values of arguments/__init__.py:66-69 (= configs/torf.json:9-12, configs/ftorf.json)
REFERENCE_ARCH = dict(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3)


def reference_network():
    """The network as the reference's ``DeformModel`` constructs it (84 encoded inputs, 522 055 parameters)."""
    return DeformNetwork(**REFERENCE_ARCH)


def flat_grad_bucket(module):
    """The gradients of the parameters that receive one, as ONE flat fp32 tensor (the reference's network:
    522 055 - 5 140 unused head values = 516 915 elements, 2.07 MB) plus the function that scatters a
    (reduced) bucket back:
    the unit of the data-parallel all-reduce over RCCL."""
    ps = [p for p in _param_list(module) if p.grad is not None]
    if not ps:
        return torch.empty(0), (lambda bucket: None)
    # the backward of this package leaves the gradients as consecutive views of one buffer: that buffer is the bucket
    g0 = ps[0].grad
    o, in_place = g0.storage_offset(), True
    for p in ps:
        g = p.grad
        if not (g.dtype == torch.float32 and g.is_contiguous() and g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr()
                and 0 <= g.storage_offset() - o < 4):
            in_place = False
            break
        o = g.storage_offset() + g.numel()
    if in_place:
        return torch.as_strided(g0, (o - g0.storage_offset(),), (1,), g0.storage_offset()), (lambda bucket: None)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])

    def scatter_back(bucket):
        o = 0
        for p in ps:
            k = p.numel()
            p.grad.copy_(bucket[o:o + k].view_as(p.grad))
            o += k
    return flat, scatter_back


def allreduce_gradients(module, dist, average=True):
    """Data-parallel step of the deformation network: every rank has rendered its own
    frame; ONE all-reduce of the flat gradient bucket (RCCL over xGMI when the process group's backend is
    ``nccl``) leaves the same summed (or averaged) gradients on all replicas.  Returns the bucket's bytes."""
    flat, scatter_back = flat_grad_bucket(module)
    if flat.numel() == 0:
        return 0
    if flat.is_cuda and dist.get_backend() != "nccl":
        # rehearsal on a CPU backend (gloo): stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    scatter_back(flat)
    return flat.numel() * flat.element_size()
