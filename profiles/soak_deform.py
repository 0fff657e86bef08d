"""Randomised soak of the deformation network against the float64 oracle: random sizes, weight / bias / head magnitudes
(from the reference's 1e-5 heads to values that leave the fp16 planes' range), shared and per-point times, sparse and dense
upstream gradients in changing order (so that the activations-on-demand policy switches back and forth).
`python profiles/soak_deform.py [seconds]` -> one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gftorf_amd import deform as D
from gftorf_amd.deform import DeformNetwork
from oracle import deform_ref

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
worst = {"forward": 0.0, "backward": 0.0}
cases = {"total": 0, "range_fallbacks": 0, "recomputed": 0, "saved": 0, "dense": 0}
t0 = time.time()
any_out_of_range = False
while time.time() - t0 < budget:
    seed = int(rng.integers(1 << 30))
    head_std = float(10 ** rng.uniform(-5, -1))
    params = deform_ref.random_params(seed, head_std=head_std)
    scale = float(10 ** rng.uniform(-0.5, 0.5))
    for k in params:
        if k.startswith("linear.") and k.endswith(".weight"):
            params[k] = (params[k] * scale ** 0.125).astype(np.float32)
        if k.endswith(".bias") and rng.random() < 0.5:
            params[k] = np.zeros_like(params[k])
    out_of_range = rng.random() < 0.15
    any_out_of_range = any_out_of_range or out_of_range
    if out_of_range:
        which = rng.integers(2)
        if which == 0:
            params["linear.%d.weight" % rng.integers(8)][rng.integers(256), rng.integers(84)] = float(rng.choice([-1, 1])) * float(rng.uniform(70, 500))
        elif which == 1:
            params["linear.%d.bias" % rng.integers(7)][rng.integers(256)] = float(rng.uniform(4500, 20000))
        cases["range_fallbacks"] += 1
    net = DeformNetwork(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3)
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    net = net.to(dev)
    n = int(rng.choice([1, 63, 64, 65, 191, 1000, 8191, 8192, 9000, 20000]))
    # a ReLU whose input is within rounding of zero may switch differently in fp32 and in float64 and changes the gradient by
    # a finite amount (tests/test_deform.py does the same): of 1.25 n candidates keep the n furthest from such an edge
    m = n + n // 4 + 4
    x = rng.random((m, 3)).astype(np.float32)
    shared_t = rng.random() < 0.5
    t = np.full((m, 1), rng.random(), np.float32) if shared_t else rng.random((m, 1)).astype(np.float32)
    margin = deform_ref.relu_margin(params, x, t)
    pick = np.sort(np.argsort(-margin)[:n])
    x, t = x[pick], t[pick]
    if margin[pick].min() < 1e-6:
        continue
    xt = torch.tensor(x, device=dev)
    tt = torch.tensor(t[:1], device=dev).expand(n, -1) if shared_t else torch.tensor(t, device=dev)
    ref = deform_ref.forward(params, x.astype(np.float64), t.astype(np.float64), dtype=np.float64)
    for it in range(int(rng.integers(1, 4))):
        frac = float(rng.choice([0.03, 0.2, 1.0]))
        keep = rng.random(n) < frac
        # (round 6: the backward kernels choose power-of-two scales from the gradients -- any magnitude, rows far apart)
        gscale = float(10 ** rng.uniform(-12, 6)) * (2.0 ** rng.integers(-20, 21, size=n) if rng.random() < 0.5 else np.ones(n))
        keep_s = keep * gscale
        g_dxyz = (rng.standard_normal((n, 3)) * keep_s[:, None]).astype(np.float32)
        g_dsh = (rng.standard_normal((n, 16, 3)) * keep_s[:, None, None]).astype(np.float32)
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xt, tt)
        e = max(rel(d_xyz.detach().cpu().numpy(), ref[0]), rel(d_sh.detach().cpu().numpy(), ref[2]))
        worst["forward_extreme" if out_of_range else "forward"] = max(worst.get("forward_extreme" if out_of_range else "forward", 0.0), e)
        torch.autograd.backward([d_xyz, d_sh], [torch.tensor(g_dxyz, device=dev), torch.tensor(g_dsh, device=dev)])
        st = D.backward_stats()
        cases["recomputed" if st["recomputed"] else "saved"] += 1
        cases["dense"] += int(st["points_processed"] == n)
        gref = deform_ref.backward(params, x.astype(np.float64), t.astype(np.float64), g_dxyz, g_dsh, dtype=np.float64)
        for name, p in net.named_parameters():
            if gref.get(name) is None:
                continue
            if not keep.any():
                eb = float(p.grad.abs().max()) if p.grad is not None else 0.0
            else:
                eb = rel(p.grad.cpu().numpy(), gref[name])
            if not out_of_range:
                worst["backward_in_range"] = max(worst.get("backward_in_range", 0.0), eb)
            if eb > worst["backward"]:
                worst["backward"], worst["where"] = eb, dict(out_of_range=bool(out_of_range), gradient_scale_max=float(np.max(gscale)), param=name, seed=seed, n=n, rows_with_gradient=int(keep.sum()), head_std=head_std,
                                                             weight_scale=scale, shared_t=bool(shared_t), stats=dict(st), ref_max=float(np.abs(gref[name]).max()))
        cases["total"] += 1
        # (a bias of 4500 .. 20000 in front of unit-size values: fp32 sums of that network carry less relative precision)
        assert e < (2e-5 if out_of_range else 3e-6) and worst["backward"] < (2e-5 if not any_out_of_range else 2e-4), (e, worst, out_of_range)
        assert worst.get("backward_in_range", 0.0) < 2e-5, worst
print(json.dumps({"seconds": round(time.time() - t0, 1), "cases": cases, "worst_forward_error_of_max_norm": worst["forward"], "worst_forward_error_networks_beyond_the_fp16_range": worst.get("forward_extreme"),
                  "worst_gradient_error_of_max_norm": worst["backward"], "worst_gradient_error_networks_inside_the_fp16_range": worst.get("backward_in_range"),
                  "backward_kernels": "bf16 planes" if os.environ.get("GFT_DEFORM_BWD_FP16") == "0" else "fp16 planes", "worst_gradient_case": worst.get("where"),
                  "tolerances": {"forward": 3e-6, "backward": 2e-5, "networks with a weight >= 70 or a bias of 4500..20000 (fp32 itself is coarser there)": {"forward": 2e-5, "backward": 2e-4}}}))
