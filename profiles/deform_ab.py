"""Deformation network forward + backward timing for A/B runs of kernel variants (environment switches pass through):
`python profiles/deform_ab.py [points]`"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
if os.environ.get("GFT_ABL_LIB"):            # ablation builds (profiles/deform_ablate.sh)
    from gftorf_amd import _lib as _l
    _l.LIB_PATH = os.path.join(ROOT, os.environ["GFT_ABL_LIB"])
from gftorf_amd import reference_network
from oracle import deform_ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
dev = torch.device("cuda:0")
params = deform_ref.random_params(9, head_std=1e-3)
net = reference_network(); net.load_state_dict({k: torch.tensor(v) for k, v in params.items()}); net = net.to(dev)
rng = np.random.default_rng(0)
x = torch.tensor(rng.random((n, 3)).astype(np.float32), device=dev)
t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
gx, gs = torch.randn((n, 3), device=dev), torch.randn((n, 16, 3), device=dev)
def timed(fn, k=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
def fwd():
    with torch.no_grad(): net(x, t)
def fwd_save():
    return net(x, t)
def both():
    d_xyz, _, d_sh, _ = net(x, t)
    torch.autograd.backward([d_xyz, d_sh], [gx, gs]); net.zero_grad(set_to_none=True)
# accuracy against the float64 oracle on a sample
ns = 4000
with torch.no_grad():
    o = net(x[:ns], t[:ns])
ref = deform_ref.forward(params, x[:ns].cpu().numpy().astype(np.float64), np.full((ns, 1), 0.4), dtype=np.float64)
e_xyz = float(np.abs(o[0].cpu().numpy() - ref[0]).max() / np.abs(ref[0]).max())
e_sh = float(np.abs(o[2].cpu().numpy() - ref[2]).max() / np.abs(ref[2]).max())
print(json.dumps({"points": n, "inference_fwd_ms": round(timed(fwd), 4), "fwd_saving_ms": round(timed(fwd_save), 4), "fwd_bwd_ms": round(timed(both), 4),
                  "err_d_xyz_vs_f64": e_xyz, "err_d_sh_vs_f64": e_sh, "env": {k: v for k, v in os.environ.items() if k.startswith("GFT_")}}))
