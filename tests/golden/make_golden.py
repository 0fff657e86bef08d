"""Generate tests/golden/*.npz from the reference's importable Python helpers.

Runs ONLY in the authoring container (needs /root/reference).  The reference
files themselves never travel; only the input/output vectors written here do.

  sh_color.npz   : utils/sh_utils.py eval_sh (lines 57-112) on random
                   coefficients/directions, degrees 0..3
  sh_phasor.npz  : the same helper on TWO channels ([P, 16, 2] coefficients: phase, amplitude) -- the polynomial of
                   forward.cu:73-125 computePhasorFromSH before its "+0.5", DC removal and amplitude clamp
  camera.npz     : utils/graphics_utils.py getProjectionMatrix (55-75),
                   getProjectionMatrixShift (77-109), getWorld2View2 (38-49)
  deform.npz     : utils/time_utils.py DeformNetwork (56-127) constructed as scene/deform_model.py:9-16 does,
                   from the defaults of arguments/__init__.py ModelParams (D 8, W 256, xyz_multires 10,
                   t_multires 10, sh_degree 3 -- also what configs/torf.json and configs/ftorf.json say):
                   forward and autograd backward on CPU (its two hard-wired ``.cuda()`` calls are made
                   no-ops for the run), with the seeded parameters of oracle/deform_ref.random_params; only
                   inputs, outputs and gradients (small ones whole, weight gradients as strided samples)
  deform_t6.npz  : the same for the class signature's defaults (t_multires 6, time_utils.py:57)
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.sh_utils import eval_sh, RGB2SH, PA2SH  # noqa: E402
from utils.graphics_utils import (getProjectionMatrix, getProjectionMatrixShift,  # noqa: E402
                                  getWorld2View2)

from utils.time_utils import DeformNetwork  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import deform_ref  # noqa: E402

DEFORM_SEED = 77


def reference_kwargs():
    """The keyword arguments scene/deform_model.py:9-15 passes, taken from ModelParams' defaults
    (arguments/__init__.py:49-69) and checked against the two shipped configs."""
    import argparse
    import json
    from arguments import ModelParams
    args = ModelParams(argparse.ArgumentParser())
    kwargs = {'D': args.D, 'W': args.W, 'xyz_multires': args.xyz_multires, 't_multires': args.t_multires,
              'sh_degree': args.sh_degree}
    for cfg in ("torf", "ftorf"):
        j = json.load(open("/root/reference/configs/%s.json" % cfg))
        assert all(j[k] == v for k, v in kwargs.items()), (cfg, kwargs)
    return kwargs


def deform_fixture(kwargs, fname):
    torch.Tensor.cuda = lambda self, *a, **k: self          # time_utils.py:121,127 on a CPU-only host
    net = DeformNetwork(**kwargs)
    net.isotropic = False
    params = deform_ref.random_params(DEFORM_SEED, t_multires=net.t_multires)
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    rng = np.random.default_rng(DEFORM_SEED + 1)
    n = 48
    x = rng.random((n, 3)).astype(np.float32)
    t = np.full((n, 1), 0.37, np.float32)
    t[n // 2:] = rng.random((n - n // 2, 1)).astype(np.float32)
    g_dxyz = rng.normal(size=(n, 3)).astype(np.float32)
    g_dsh = rng.normal(size=(n, 16, 3)).astype(np.float32)
    d_xyz, d_rot, d_sh, d_sh_p = net(torch.tensor(x), torch.tensor(t))
    ((d_xyz * torch.tensor(g_dxyz)).sum() + (d_sh * torch.tensor(g_dsh)).sum()).backward()
    out = dict(x=x, t=t, g_dxyz=g_dxyz, g_dsh=g_dsh, seed=np.int64(DEFORM_SEED),
               d_xyz=d_xyz.detach().numpy(), d_rot=d_rot.numpy(), d_sh=d_sh.detach().numpy(),
               d_sh_p=d_sh_p.numpy())
    none = []
    for name, p in net.named_parameters():
        if p.grad is None:
            none.append(name)
        elif p.grad.numel() <= 4096:
            out["grad:" + name] = p.grad.numpy()
        else:
            out["grad_s:" + name] = p.grad.numpy()[::8, ::4]   # strided sample of a [256, in] matrix
    out["grad_none"] = np.array(none)
    out["param_names"] = np.array(list(net.state_dict().keys()))
    out["param_shapes"] = np.array([";".join(map(str, v.shape)) for v in net.state_dict().values()])
    out["num_params"] = np.int64(sum(p.numel() for p in net.parameters()))
    out["kwargs"] = np.array([net.D, net.W, net.xyz_multires, net.t_multires, int(round(net.num_shs ** 0.5)) - 1], np.int64)
    np.savez(os.path.join(HERE, fname), **out)


def main():
    rng = np.random.default_rng(20251003)
    n = 64
    sh = rng.normal(0, 0.5, (n, 16, 3)).astype(np.float32)       # kernel layout [coeff][channel]
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {}
    for deg in range(4):
        # helper wants [..., C, coeff]
        res = eval_sh(deg, torch.tensor(sh).permute(0, 2, 1), torch.tensor(dirs))
        out[f"deg{deg}"] = res.numpy().astype(np.float32)
    x = rng.random((8, 3)).astype(np.float32)
    np.savez(os.path.join(HERE, "sh_color.npz"), sh=sh, dirs=dirs,
             rgb2sh_in=x, rgb2sh_out=RGB2SH(torch.tensor(x)).numpy(),
             pa2sh_out=PA2SH(torch.tensor(x)).numpy(), **out)

    # computePhasorFromSH (forward.cu:73-125) is eval_sh on two channels; amplitude DCs on both sides of the clamp
    sh_p = rng.normal(0, 0.3, (n, 16, 2)).astype(np.float32)
    sh_p[:, 0, 1] = rng.uniform(-2.5, 1.5, n).astype(np.float32)
    outp = {}
    for deg in range(4):
        outp[f"deg{deg}"] = eval_sh(deg, torch.tensor(sh_p).permute(0, 2, 1), torch.tensor(dirs)).numpy().astype(np.float32)
    np.savez(os.path.join(HERE, "sh_phasor.npz"), sh_p=sh_p, dirs=dirs, **outp)

    cams = {}
    for name, (W, H, fovx_deg, zn, zf) in {
            "bench640": (640, 480, 60.0, 0.45, 6.05),
            "bench1080": (1920, 1080, 60.0, 0.45, 6.05),
            "c1_256": (256, 256, 60.0, 0.45, 6.05)}.items():
        tanx = math.tan(math.radians(fovx_deg) * 0.5)
        tany = tanx * H / W
        fovx, fovy = 2 * math.atan(tanx), 2 * math.atan(tany)
        cams[name + "_proj"] = getProjectionMatrix(zn, zf, fovx, fovy).numpy()
        cams[name + "_args"] = np.array([W, H, fovx, fovy, zn, zf], np.float64)
    W, H, fx, fy, cx, cy = 320, 240, 260.0, 262.0, 155.5, 118.25
    fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
    cams["shift_proj"] = getProjectionMatrixShift(0.45, 6.05, fx, fy, cx, cy, W, H, fovx, fovy).numpy()
    cams["shift_args"] = np.array([0.45, 6.05, fx, fy, cx, cy, W, H, fovx, fovy], np.float64)
    # world-to-view for a non-trivial pose
    a = 0.3
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    t = np.array([0.1, -0.2, 0.3])
    cams["w2v_R"], cams["w2v_t"] = R, t
    cams["w2v"] = getWorld2View2(R, t)
    np.savez(os.path.join(HERE, "camera.npz"), **cams)
    kw = reference_kwargs()
    assert kw == dict(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3), kw
    deform_fixture(kw, "deform.npz")
    deform_fixture({}, "deform_t6.npz")
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
