"""BASELINE config 3's loop shape (bench_loop.py: deform query, activations, input assembly, colour + ToF rasterizer call,
ToF loss, backward, densification statistics, Adam on the Gaussians and the network) with its device work captured in a
HIP graph per (SH degree, network on / off) against the same loop run eagerly: the same statements on the same values, so
after N iterations the parameters agree to what the order of the float atomics (rasterizer backward) and of the
network's weight-gradient sums leaves open."""
import sys
import os

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
def test_captured_iterations_equal_eager_ones(gpu):
    import bench_loop
    from gftorf_amd import api
    cfg = dict(bench_loop.C3, P=20_000, W=128, H=96, views=5, warm_up=9)
    N = 22                       # eager 1..3, replays 4..9 (network off), eager 10..12, replays 13..22 (network on)
    runs = {}
    for graph in (False, True):
        api._instance_hint.clear()
        torch.manual_seed(0)
        iteration, info = bench_loop.build_loop(gpu, cfg, seed=77, graph=graph)
        losses = [iteration(it).clone() for it in range(1, N + 1)]
        torch.cuda.synchronize()
        runs[graph] = dict(losses=[float(l) for l in losses], par={k: v.detach().clone() for k, v in info["par"].items()},
                           net=[p.detach().clone() for p in info["net"].parameters()], graphs=info["graphs"]())
    assert runs[True]["graphs"] == {(0, False): True, (0, True): True}          # both configurations were captured
    # the first iterations are the same eager code on the same values: equal losses to the atomics' order
    np.testing.assert_allclose(runs[True]["losses"][:3], runs[False]["losses"][:3], rtol=1e-5)
    # ... and the replayed ones follow the eager run (the loss moves by ~30 % over the run; a replay that read stale inputs
    # -- another view's camera, an old background, a learning rate that stood still -- would not)
    np.testing.assert_allclose(runs[True]["losses"], runs[False]["losses"], rtol=2e-3)
    assert runs[False]["losses"][-1] < 0.9 * runs[False]["losses"][0]
    fresh = bench_loop.build_loop(gpu, cfg, seed=77, graph=False)[1]
    start = {k: v.detach() for k, v in fresh["par"].items()}
    for i, (n, p0) in enumerate(fresh["net"].named_parameters()):
        start["net." + n] = p0.detach()
        runs[False]["par"]["net." + n], runs[True]["par"]["net." + n] = runs[False]["net"][i], runs[True]["net"][i]
    report = {}
    for k, ref in runs[False]["par"].items():
        moved = float((ref - start[k]).abs().max())
        err = (runs[True]["par"][k] - ref).abs()
        if moved == 0.0:         # (no loss reaches it: the colour features with lambda_color = 0, SH rest coefficients at degree 0,
            assert float(err.max()) == 0.0, k          # the network's unused rot / a heads)
            continue
        # (Adam turns the sign of a gradient into a step of the learning rate: where a gradient is rounding noise around
        # zero -- a Gaussian hardly any pixel sees, a weight of the network's near-zero heads -- two orders of the same
        # float sums can walk apart by steps, so the bound is on the mean and on the share of elements that differ
        # visibly, not on the worst element)
        report[k] = (round(float(err.mean()) / moved, 5), round(float((err > 0.05 * moved).float().mean()), 5), round(float(err.max()) / moved, 4))
        if os.environ.get("GFT_TEST_REPORT_ONLY"):
            continue
        if k.startswith("net.linear"):
            # the trunk's gradients reach it through heads of magnitude 1e-5 (time_utils.py:83-101) and Adam normalises them:
            # TWO EAGER runs of this loop differ by 0.025 (mean) / 12 % of the elements here (the rasterizer's atomics),
            # graph against eager was measured at 0.05 / 40 %; a stale learning rate or a step that does not land gives > 0.5
            assert report[k][0] <= 0.15, (k, report[k])
        else:
            assert report[k][0] <= 5e-3 and report[k][1] <= 2e-2, (k, report[k])
    print("graph vs eager, per parameter (mean error, share above 5 %, worst) in units of the largest move:", report)
