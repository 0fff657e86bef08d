import os, sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from gftorf_amd import reference_network
from oracle import deform_ref
dev = torch.device("cuda:0")
res = {}
for head_std, tag in ((1e-3, "test"), (None, "ref_init")):
    params = deform_ref.random_params(9, head_std=head_std) if head_std else None
    net = reference_network()
    if params: net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    net = net.to(dev)
    for n in (300_000, 100_000, 20_000):
        x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
        def fwd():
            with torch.no_grad(): return net(x, t)
        for _ in range(5): o = fwd()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fwd()
        b.record(); torch.cuda.synchronize()
        res["%s_%d" % (tag, n)] = round(a.elapsed_time(b) / 20, 4)
        res["%s_%d_sum" % (tag, n)] = float(o[2].double().abs().sum())
print(json.dumps({"light": os.environ.get("GFT_EXP_LIGHT", "0"), **res}))
