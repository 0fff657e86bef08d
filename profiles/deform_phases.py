"""Where a workgroup of the deformation network's forward spends its time: needs the -DDF_ABL=16 build
(profiles/deform_ablate.sh build), which writes s_memtime stamps of every wave into d_sh instead of the results.
`GFT_ABL_LIB=gftorf_amd/_abl/lib_16.so python profiles/deform_phases.py`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gftorf_amd import _lib as _l
_l.LIB_PATH = os.path.join(ROOT, os.environ["GFT_ABL_LIB"])
from gftorf_amd import reference_network
from oracle import deform_ref
n = 300_000
dev = torch.device("cuda:0")
params = deform_ref.random_params(9, head_std=1e-3)
net = reference_network(); net.load_state_dict({k: torch.tensor(v) for k, v in params.items()}); net = net.to(dev)
x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
names = ["encode", "bar0"] + sum([["gemm%d" % l, "barA%d" % l, "epi%d" % l, "barB%d" % l] for l in range(8)], []) + ["heads"]
for label, nw, ctxm in (("saving (4 waves x 64 columns, two workgroups per CU)", 4, torch.enable_grad), ("inference (8 waves x 32 columns)", 8, torch.no_grad)):
    for _ in range(3):
        with ctxm():
            o = net(x, t)
    torch.cuda.synchronize()
    wgs = n // 64
    st = o[2].detach().reshape(-1)[:wgs * 64 * 48].view(torch.int64).view(wgs, 64 * 24)[:, :nw * 48].reshape(wgs, nw, 48)[:, :, :len(names) + 1].cpu().numpy()
    d = np.diff(st, axis=2).astype(np.float64)                    # [wg][wave][phase]
    life = (st[:, :, -1] - st[:, :, 0]).astype(np.float64)
    span = float(st[:, :, -1].max() - st[:, :, 0].min())
    print(label)
    print("  kernel span %.0f ticks; mean workgroup life %.0f ticks (%.1f %% of the span; %.2f workgroup lives per span)" % (span, life.mean(), 100 * life.mean() / span, span / life.mean()))
    tot = d.mean(axis=(0, 1))
    grp = {}
    for nm, v in zip(names, tot):
        key = nm.rstrip("0123456789")
        grp[key] = grp.get(key, 0.0) + v
    for k, v in grp.items():
        print("  %-8s %8.0f ticks  %5.1f %%" % (k, v, 100 * v / life.mean()))
    print("  per layer gemm:", [int(v) for nm, v in zip(names, tot) if nm.startswith("gemm")])
    print("  per layer epi :", [int(v) for nm, v in zip(names, tot) if nm.startswith("epi")])
    print("  per layer barA:", [int(v) for nm, v in zip(names, tot) if nm.startswith("barA")])
    print("  per layer barB:", [int(v) for nm, v in zip(names, tot) if nm.startswith("barB")])
    if nw == 4:
        lb = o[2].detach().reshape(-1)[:wgs * 64 * 48].view(torch.int64).view(wgs, 64 * 24)[:, 47].cpu().numpy()
        print("  LDS_BASE values of the first 1024 workgroups:", dict(zip(*np.unique(lb[:1024], return_counts=True))))
