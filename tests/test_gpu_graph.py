"""The forward without a host read (gft_forward_enqueue) and an iteration captured in a graph: a C3-shaped pair of
rasterizer calls (100 k Gaussians at 320 x 240, two cameras on the same Gaussians) with its backward, replayed by
torch.cuda.CUDAGraph, against the same calls run eagerly through the blocking flow -- every output bit for bit, gradients
to the order of their float atomics (which is undefined eagerly too; the deterministic test mode is not capturable)."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu


def _leaves(scene, dev):
    g = scene["gaussians"]
    leaf = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((g["means3D"].shape[0], 3), device=dev, requires_grad=True)
    return leaf, m2


def _render(rast, leaf, m2, scene):
    return rast(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])


def test_forward_without_a_host_read(oracle, gpu):
    """gftorf_amd.api.no_host_read: the same kernels queued without waiting for the instance count -- outputs and gradients
    equal the blocking flow's, the device's posting is found afterwards; a frame that does not fit its buffer is reported by
    the next call of the shape, and the buffer grows."""
    from gftorf_amd import api
    scene = Hh.small_scene(P=6000, W=96, H=64, seed=31, scale_lo=0.02, scale_hi=0.1)
    f, b = Hh.run_oracle(oracle, scene)
    api._instance_hint.clear()
    ref_out, ref_grads, _ = Hh.run_gpu(scene, gpu)                  # blocking: first frame of the shape, two stages
    Hh.run_gpu(scene, gpu)                                          # ... and the one-call flow
    keep = api.no_host_read
    api.no_host_read = True
    try:
        for _ in range(2):
            out, grads, _ = Hh.run_gpu(scene, gpu)
            assert api.last_call_stats["num_rendered"] == -1        # nobody told the host
            for k in ref_out:
                np.testing.assert_array_equal(out[k], ref_out[k], err_msg=k)
            for k in ref_grads:
                if ref_grads[k] is not None:
                    Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=1e-5)
        st = [x for x in api.enqueue_status() if x["key"][1:4] == (6000, 96, 64)]
        assert len(st) == 1 and st[0]["posted"] and st[0]["num_rendered"] == f.num_rendered and not st[0]["overflow"]
        # a buffer that is too small: the frame's outputs are undefined (and its backward walks nothing), the next call says so,
        # the one after is right again
        key = st[0]["key"]
        api._status[key]["np"][3] = 0                               # (the last posting would correct the guess below)
        api._instance_hint[key] = (64, 0)
        Hh.run_gpu(scene, gpu)
        torch.cuda.synchronize()
        assert [x for x in api.enqueue_status() if x["key"] == key][0]["overflow"]
        with pytest.raises(RuntimeError, match="outputs were undefined"):
            Hh.run_gpu(scene, gpu)
        for _ in range(3):                                          # (whatever buffers of the failed frame come round)
            out, grads, _ = Hh.run_gpu(scene, gpu)
            for k in ref_out:
                np.testing.assert_array_equal(out[k], ref_out[k], err_msg=k)
            for k in ref_grads:
                if ref_grads[k] is not None:
                    Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=1e-5)
        # A host that runs AHEAD of the device: frame N overflows, the host has not seen its posting when it queues frame
        # N + 1 (which fits and overwrites the posting), and looks for the first time in front of frame N + 2.  The overflow
        # count is a word of the status block that nothing clears: frame N is still reported.
        st = api._status[key]
        good_hint = api._instance_hint[key]
        st["np"][3] = 0                                             # (the last posting would correct the guess below)
        api._instance_hint[key] = (64, 0)
        Hh.run_gpu(scene, gpu)                                      # frame N: does not fit
        st["np"][3] = 0
        st["np"][13] = st["seen"]                                   # (as if the copy of its posting had not arrived yet)
        api._instance_hint[key] = good_hint
        Hh.run_gpu(scene, gpu)                                      # frame N + 1: fits, its posting replaces frame N's
        torch.cuda.synchronize()
        rep = [x for x in api.enqueue_status() if x["key"] == key][0]
        assert not rep["overflow"] and rep["overflows"] == 2 and rep["max_overflow_instances"] == f.num_rendered
        with pytest.raises(RuntimeError, match="outputs were undefined"):
            Hh.run_gpu(scene, gpu)                                  # frame N + 2 finds the count
        out, grads, _ = Hh.run_gpu(scene, gpu)
        for k in ref_out:
            np.testing.assert_array_equal(out[k], ref_out[k], err_msg=k)
        # ... and a shape the operator has never seen (the first frame of a process, a new P after a densification step)
        # takes the blocking flow once instead of raising, then goes on without host reads
        other = Hh.small_scene(P=5000, W=96, H=64, seed=32, scale_lo=0.02, scale_hi=0.1)
        fo, _bo = Hh.run_oracle(oracle, other)
        first, _g, _ = Hh.run_gpu(other, gpu)
        assert api.last_call_stats["num_rendered"] == fo.num_rendered
        second, _g, _ = Hh.run_gpu(other, gpu)
        assert api.last_call_stats["num_rendered"] == -1
        for k in first:
            np.testing.assert_array_equal(first[k], second[k], err_msg=k)
    finally:
        api.no_host_read = keep
        api._instance_hint.clear()


def test_graph_captured_iteration_equals_the_eager_one(gpu):
    from gftorf_amd import api, synth, GaussianRasterizer
    P, W, H = 100_000, 320, 240
    cams = [synth.look_at_w2c(0.05, -0.02, 0.0, (0.05, 0.0, 0.1)), synth.look_at_w2c(-0.08, 0.03, 0.01, (-0.1, 0.02, 0.15))]
    scenes = [Hh.small_scene(P=P, W=W, H=H, seed=41, scale_lo=0.004, scale_hi=0.03, opacity=0.1, w2c=c) for c in cams]
    leaf, m2 = _leaves(scenes[0], gpu)
    rasts = [GaussianRasterizer(raster_settings=Hh.gpu_settings(sc, gpu)) for sc in scenes]
    ups = [[torch.tensor(sc["grads"][k], device=gpu) for k in Hh.GRAD_KEYS] for sc in scenes]

    def iteration():
        outs = [_render(r, leaf, m2, sc) for r, sc in zip(rasts, scenes)]
        diff = [t for o in outs for t in (o[0], o[1], o[2], o[4], o[6])]
        torch.autograd.backward(diff, [u for up in ups for u in up])
        return outs

    def clear():
        for v in leaf.values():
            v.grad = None
        m2.grad = None

    def eager():
        clear()
        outs = iteration()
        torch.cuda.synchronize()
        # (DETACHED copies: on this image -- torch 2.10 + ROCm 7.0 -- capturing a backward while ANY other autograd graph on
        # the same leaves is alive, even a pure-torch one, ends in a segmentation fault inside capture_end
        # (profiles/experiments/graph_capture_live_autograd_graph_repro.py shows it without this package))
        return ([[t.detach().clone() for t in o] for o in outs], {k: v.grad.clone() for k, v in leaf.items()}, m2.grad.clone())

    def churn():
        # memory of the eager pools handed out, poisoned and freed again: a graph that had baked in a buffer of those pools
        # (kept accumulators, kept gradient tensors) would read or spoil it at the next replay
        for n in (300_000_000, 40_000_000, 1_000_000):
            x = torch.full((n,), float("nan"), device=gpu)
            del x

    def compare(static_outs, static_grads, ref_outs, ref_grads, ref_m2, what):
        for o, r in zip(static_outs, ref_outs):
            for a, e in zip(o, r):
                assert torch.equal(a, e), what
        for k in leaf:
            Hh.assert_close("%s %s" % (what, k), ref_grads[k].cpu().numpy(), static_grads[k].cpu().numpy(), rtol_max=1e-5)
        Hh.assert_close(what + " means2D", ref_m2.cpu().numpy(), static_grads["means2D"].cpu().numpy(), rtol_max=1e-5)

    api._instance_hint.clear()
    try:
        eager()                                             # first frame of the shape: two-stage flow
        ref_outs, ref_grads, ref_m2 = eager()
        # warm-up on a side stream, then the capture (torch's recipe for whole-iteration graphs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                clear()
                iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        clear()
        churn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_outs = iteration()
        # (the gradient tensors the graph writes: the leaves' .grad as the capture left them)
        static_grads = dict({k: v.grad for k, v in leaf.items()}, means2D=m2.grad)
        for rep in range(4):
            graph.replay()
            torch.cuda.synchronize()
            compare(static_outs, static_grads, ref_outs, ref_grads, ref_m2, "replay %d" % rep)
            churn()
            if rep == 1:
                eager()                                     # an eager frame of the same shape between two replays
        st = [x for x in api.enqueue_status() if x["key"][1:4] == (P, W, H)]
        assert st and all(x["posted"] and not x["overflow"] for x in st)
    finally:
        api._instance_hint.clear()


def test_graph_replay_follows_its_inputs(gpu):
    """Other values in the captured tensors: the replay renders the new frame (outputs bit for bit against an eager render
    of the same values)."""
    from gftorf_amd import api, GaussianRasterizer
    P, W, H = 30_000, 200, 150
    scene = Hh.small_scene(P=P, W=W, H=H, seed=43, scale_lo=0.005, scale_hi=0.04)
    leaf, m2 = _leaves(scene, gpu)
    rast = GaussianRasterizer(raster_settings=Hh.gpu_settings(scene, gpu))
    ups = [torch.tensor(scene["grads"][k], device=gpu) for k in Hh.GRAD_KEYS]
    static_grads = {}

    def iteration():
        o = _render(rast, leaf, m2, scene)
        torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
        return o

    def clear():
        for v in leaf.values():
            v.grad = None
        m2.grad = None

    api._instance_hint.clear()
    try:
        for _ in range(2):
            clear()
            iteration()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            clear()
            iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        clear()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = iteration()
        static_grads = {k: v.grad for k, v in leaf.items()}
        rng = np.random.default_rng(3)
        for step in range(3):
            with torch.no_grad():
                leaf["means3D"].add_(torch.tensor(rng.normal(0, 0.01, (P, 3)), dtype=torch.float32, device=gpu))
                leaf["opacities"].mul_(0.95)
            graph.replay()
            torch.cuda.synchronize()
            got = [t.detach().clone() for t in static_out]
            got_g = {k: v.clone() for k, v in static_grads.items()}
            clear()
            ref = iteration()
            torch.cuda.synchronize()
            for a, e in zip(got, ref):
                assert torch.equal(a, e.detach()), step
            for k in leaf:
                Hh.assert_close("step %d %s" % (step, k), leaf[k].grad.cpu().numpy(), got_g[k].cpu().numpy(), rtol_max=1e-5)
            del ref
            # (the eager iteration gave the leaves new .grad tensors; the graph keeps writing its own)
            for k, v in leaf.items():
                v.grad = None
        st = [x for x in api.enqueue_status() if x["key"][1:4] == (P, W, H)]
        assert st and not st[0]["overflow"]
    finally:
        api._instance_hint.clear()


def test_two_deterministic_backwards_in_one_graph(gpu):
    """The deterministic backward (fixed-order sums, bit-reproducible) inside a captured iteration of two views: every replay
    equals the eager deterministic run bit for bit.  Round 5 had parked this as "wrong sums from the second replay on"; the
    cause was not in the mode: the library cleared the mode's partial rows with hipMemsetAsync, and a large memset node in
    a replayed graph is not ordered against its neighbours on this platform (ROCm 7.0 / torch 2.10) -- the first replay met
    fresh zero pages and was right, later ones were not.  The library clears with a kernel now (gft_zero_async); with the
    memset in place this test fails at `replay 1`."""
    from gftorf_amd import api, synth, GaussianRasterizer
    P, W, H = 20_000, 128, 96
    cams = [synth.look_at_w2c(0.05, -0.02, 0.0, (0.05, 0.0, 0.1)), synth.look_at_w2c(-0.08, 0.03, 0.01, (-0.1, 0.02, 0.15))]
    scenes = [Hh.small_scene(P=P, W=W, H=H, seed=41, scale_lo=0.004, scale_hi=0.03, opacity=0.1, w2c=c) for c in cams]
    leaf, m2 = _leaves(scenes[0], gpu)
    rasts = [GaussianRasterizer(raster_settings=Hh.gpu_settings(sc, gpu)) for sc in scenes]
    ups = [[torch.tensor(sc["grads"][k], device=gpu) for k in Hh.GRAD_KEYS] for sc in scenes]

    def iteration():
        outs = [_render(r, leaf, m2, sc) for r, sc in zip(rasts, scenes)]
        torch.autograd.backward([t for o in outs for t in (o[0], o[1], o[2], o[4], o[6])], [u for up in ups for u in up])

    def clear():
        for v in leaf.values():
            v.grad = None
        m2.grad = None

    def eager():
        clear()
        iteration()
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in leaf.items()}

    keep = api._DETERMINISTIC
    api._DETERMINISTIC = True
    api._instance_hint.clear()
    try:
        eager()
        ref, again = eager(), eager()
        for k in ref:
            assert torch.equal(ref[k], again[k]), k                  # the mode's promise, eagerly
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            clear()
            iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        clear()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            iteration()
        static = {k: v.grad for k, v in leaf.items()}
        for rep in range(4):
            graph.replay()
            torch.cuda.synchronize()
            for k in ref:
                assert torch.equal(static[k], ref[k]), "replay %d: %s" % (rep, k)
    finally:
        api._DETERMINISTIC = keep
        api._instance_hint.clear()
