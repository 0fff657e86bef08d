import os, sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from oracle import deform_ref
from test_deform import _net, _inputs, _rel
dev = torch.device("cuda:0")
net, params = _net(12, dev)
n = 5000
x, t = _inputs(n + n // 8, 200 + n, False)
keep = np.sort(np.argsort(-deform_ref.relu_margin(params, x, t))[:n]); x, t = x[keep], t[keep]
rng = np.random.default_rng(n)
g_dxyz, g_dsh = rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 16, 3)).astype(np.float32)
d_xyz, _, d_sh, _ = net(torch.tensor(x, device=dev), torch.tensor(t, device=dev))
((d_xyz * torch.tensor(g_dxyz, device=dev)).sum() + (d_sh * torch.tensor(g_dsh, device=dev)).sum()).backward()
ref = deform_ref.backward(params, x, t, g_dxyz, g_dsh, dtype=np.float64)
ref32 = deform_ref.backward(params, x, t, g_dxyz, g_dsh)
out = {}
for name, p in net.named_parameters():
    if ref[name] is not None:
        out[name] = (_rel(p.grad.cpu().numpy(), ref[name]), _rel(ref32[name], ref[name]))
print(json.dumps({"env": os.environ.get("GFT_DEFORM_BWD_FP16", "1"), "worst_device": max(v[0] for v in out.values()), "worst_numpy_fp32": max(v[1] for v in out.values()),
                  "linear": {k: ["%.2e" % a for a in v] for k, v in out.items() if "weight" in k and "linear" in k}}))
