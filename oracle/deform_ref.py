"""ORACLE (test infrastructure only) of the deformation network -- SURVEY section 8(f) row 2.

CPU restatement (numpy) of ``utils/time_utils.py`` of the reference:

* :func:`embed`        -- ``get_embedder`` / ``Embedder.embed`` (time_utils.py:8-53): include_input,
  log-sampled frequencies 2^0 .. 2^(multires-1), per frequency sin then cos of all input dims.
* :func:`forward`      -- ``DeformNetwork.forward`` (time_utils.py:103-127): 8 x (Linear + ReLU) of width
  256, the embedded input concatenated IN FRONT of the activations after layer 4 (``skips = [D // 2]``),
  heads ``xyz_warp`` and ``r``/``g``/``b`` (stacked on the last axis); ``d_rot`` and ``d_sh_p`` are
  returned as ZEROS by the reference (time_utils.py:127), the ``rot`` and ``a`` heads never reach an output.
* :func:`backward`     -- the adjoint of the above, written out by hand (chain rule of Linear / ReLU);
  inputs are not differentiated (the reference detaches them, scene/gaussian_model.py:172).

Parameters are a dict with the reference's ``state_dict`` names (``linear.0.weight`` ... ``b.bias``), numpy
arrays in torch's ``[out, in]`` layout.  ``dtype=np.float64`` gives the accuracy reference of the float32
comparisons.  Pinned against the reference's own module by ``tests/golden/deform.npz``
(``tests/golden/make_golden.py``).

* :func:`deform_eager` -- the same statements on torch tensors (eager ``F.linear`` / ``relu`` / ``cat``, autograd
  for the backward): what the reference's module runs on a GPU; bench.py times it beside the HIP path.

Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may import this module.
"""
import numpy as np

D, W, XYZ_MULTIRES, NUM_SHS = 8, 256, 10, 16
# The reference constructs the network with t_multires = 10 (scene/deform_model.py:9-16 from
# arguments/__init__.py:66-69, configs/torf.json:9-12, configs/ftorf.json): 84 encoded inputs.  The class
# signature's default is 6 (time_utils.py:57): 76.  Every function takes ``t_multires``.
T_MULTIRES = 10
T_MULTIRES_CLASS_DEFAULT = 6
SKIP = D // 2                                    # time_utils.py:62
XYZ_CH = 3 + 3 * 2 * XYZ_MULTIRES                # 63


def in_ch(t_multires=T_MULTIRES, xyz_multires=XYZ_MULTIRES):
    """Columns of cat([x_emb, t_emb]) (time_utils.py:64-65): 84 for (10, 10), 76 for (10, 6)."""
    return 3 + 6 * xyz_multires + 1 + 2 * t_multires


IN_CH = in_ch()                                  # 84
HEADS = ("xyz_warp", "r", "g", "b")              # heads that reach an output
UNUSED = ("rot", "a")                            # computed by the reference, discarded (time_utils.py:118-127)


def param_shapes(t_multires=T_MULTIRES, xyz_multires=XYZ_MULTIRES):
    """state_dict names -> shapes (time_utils.py:68-81)."""
    s = {}
    n_in = in_ch(t_multires, xyz_multires)
    for i in range(D):
        fan_in = n_in if i == 0 else (W + n_in if i == SKIP + 1 else W)
        s["linear.%d.weight" % i] = (W, fan_in)
        s["linear.%d.bias" % i] = (W,)
    for name, out in (("xyz_warp", 3), ("rot", 4), ("r", NUM_SHS), ("g", NUM_SHS), ("b", NUM_SHS), ("a", NUM_SHS)):
        s[name + ".weight"] = (out, W)
        s[name + ".bias"] = (out,)
    return s


def arch_of(params):
    """(xyz_multires, t_multires) of a parameter dict, from the width of ``linear.0.weight`` (xyz_multires is 10
    in every configuration of the reference)."""
    n_in = params["linear.0.weight"].shape[1]
    t2 = n_in - XYZ_CH - 1
    assert t2 >= 0 and t2 % 2 == 0, n_in
    return XYZ_MULTIRES, t2 // 2


def random_params(seed, head_std=0.05, t_multires=T_MULTIRES):
    """Seeded parameters (synthetic data: the generator lives with the other synthetic inputs in ``gftorf_amd.synth``, so that
    the benches' HIP legs need nothing from this package; re-exported here for the tests)."""
    from gftorf_amd import synth
    p = synth.random_deform_params(seed, head_std=head_std, t_multires=t_multires)
    assert {k: v.shape for k, v in p.items()} == param_shapes(t_multires)
    return p


def embed_one(v, multires, dtype):
    """Embedder.embed (time_utils.py:24-53) of v[n, d]."""
    v = np.asarray(v, dtype)
    parts = [v]
    for f in range(multires):
        freq = dtype(2.0 ** f)                   # 2 ** linspace(0, multires-1, multires), exact
        parts.append(np.sin(v * freq))
        parts.append(np.cos(v * freq))
    return np.concatenate(parts, axis=-1)


def embed(x, t, dtype=np.float32, t_multires=T_MULTIRES):
    """cat([x_emb, t_emb]) (time_utils.py:104-107): [n, 63 + 1 + 2 t_multires]."""
    return np.concatenate([embed_one(x, XYZ_MULTIRES, dtype), embed_one(t, t_multires, dtype)], axis=-1)


def _trunk(params, x, t, dtype):
    emb = embed(x, t, dtype, arch_of(params)[1])
    h = emb
    inputs, outs = [], []
    for i in range(D):                           # time_utils.py:109-113
        w = params["linear.%d.weight" % i].astype(dtype)
        b = params["linear.%d.bias" % i].astype(dtype)
        inputs.append(h)
        z = h @ w.T + b
        a = np.maximum(z, 0)
        outs.append(a)
        h = np.concatenate([emb, a], axis=-1) if i == SKIP else a
    return emb, inputs, outs, h


def relu_margin(params, x, t):
    """Per point, the smallest |pre-activation| over all 8 x 256 units (float64).  The gradient is
    discontinuous where a pre-activation crosses zero, so comparisons of gradients between two
    arithmetics leave out the points whose margin is within rounding distance of that edge."""
    emb = embed(x, t, np.float64, arch_of(params)[1])
    h = emb
    margin = np.full(emb.shape[0], np.inf)
    for i in range(D):
        z = h @ params["linear.%d.weight" % i].astype(np.float64).T + params["linear.%d.bias" % i].astype(np.float64)
        margin = np.minimum(margin, np.abs(z).min(axis=1))
        a = np.maximum(z, 0)
        h = np.concatenate([emb, a], axis=-1) if i == SKIP else a
    return margin


def forward(params, x, t, dtype=np.float32):
    """Returns (d_xyz[n,3], d_rot[n,4] = 0, d_sh[n,16,3], d_sh_p[n,16,2] = 0) (time_utils.py:115-127)."""
    _, _, _, h = _trunk(params, x, t, dtype)
    head = lambda n: h @ params[n + ".weight"].astype(dtype).T + params[n + ".bias"].astype(dtype)
    d_xyz = head("xyz_warp")
    d_sh = np.stack([head("r"), head("g"), head("b")], axis=-1)
    n = d_xyz.shape[0]
    return d_xyz, np.zeros((n, 4), dtype), d_sh, np.zeros((n, NUM_SHS, 2), dtype)


def backward(params, x, t, g_dxyz, g_dsh, dtype=np.float32):
    """Gradients of sum(d_xyz * g_dxyz) + sum(d_sh * g_dsh) w.r.t. every parameter that reaches an output;
    the ``rot`` / ``a`` heads get None as under the reference's autograd."""
    emb, inputs, outs, h = _trunk(params, x, t, dtype)
    g = {n + s: None for n in UNUSED for s in (".weight", ".bias")}
    g_dxyz = np.asarray(g_dxyz, dtype)
    g_dsh = np.asarray(g_dsh, dtype)
    dh = np.zeros_like(h)
    for name, go in (("xyz_warp", g_dxyz), ("r", g_dsh[:, :, 0]), ("g", g_dsh[:, :, 1]), ("b", g_dsh[:, :, 2])):
        g[name + ".weight"] = go.T @ h
        g[name + ".bias"] = go.sum(axis=0)
        dh = dh + go @ params[name + ".weight"].astype(dtype)
    for i in reversed(range(D)):
        dz = dh * (outs[i] > 0)
        g["linear.%d.weight" % i] = dz.T @ inputs[i]
        g["linear.%d.bias" % i] = dz.sum(axis=0)
        dh = dz @ params["linear.%d.weight" % i].astype(dtype)
        if i == SKIP + 1:
            dh = dh[:, emb.shape[1]:]            # the embedded part of the skip input has no parameters
    return g


def deform_eager(params, x, t):
    """time_utils.py:103-127 on torch tensors (``params``: name -> tensor, torch layout), including the two
    heads the reference computes and throws away."""
    import torch
    import torch.nn.functional as F

    def emb_one(v, multires):
        parts = [v]
        for f in range(multires):
            parts += [torch.sin(v * float(2 ** f)), torch.cos(v * float(2 ** f))]
        return torch.cat(parts, -1)

    t_multires = (params["linear.0.weight"].shape[1] - XYZ_CH - 1) // 2
    emb = torch.cat([emb_one(x, XYZ_MULTIRES), emb_one(t, t_multires)], dim=-1)
    h = emb
    for i in range(D):
        h = F.relu(F.linear(h, params["linear.%d.weight" % i], params["linear.%d.bias" % i]))
        if i == SKIP:
            h = torch.cat([emb, h], -1)
    head = lambda n: F.linear(h, params[n + ".weight"], params[n + ".bias"])
    d_xyz = head("xyz_warp")
    d_sh_a = head("a")                                   # time_utils.py:119, unused
    d_rot = head("rot")                                  # time_utils.py:125, replaced by zeros
    d_sh = torch.stack([head("r"), head("g"), head("b")], dim=-1)
    return d_xyz, torch.zeros_like(d_rot), d_sh, torch.zeros(d_sh_a.shape + (2,), device=x.device)
