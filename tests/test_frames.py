"""BASELINE.json config 4's per-rank step (gftorf_amd.frames.FrameStep) on the GPU: the composed HIP step against the
same step composed from the oracles, and the path's one collective through a real RCCL communicator."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_composed_frame_step_matches_composed_oracles(oracle, gpu):
    """deform query at the frame's time -> input assembly -> rasterizer forward + backward -> network backward, once through
    the HIP kernels and once through the host-side oracles (eager network, eager assembly, C oracle of the rasterizer)
    composed by the same FrameStep: images, Gaussian gradients and network gradients agree, for two different frames."""
    from gftorf_amd import GaussianRasterizer, reference_network
    from gftorf_amd.frames import FrameStep
    from test_dist_gloo import _frame_step_parts
    scene, cpu_net, cpu_leaf, mask, upstream, cpu_render, cpu_assemble = _frame_step_parts(P=800, W=96, H=64)
    net = reference_network()
    net.load_state_dict(cpu_net.state_dict())
    net = net.to(gpu)
    leaf = {k: v.detach().to(gpu).requires_grad_(True) for k, v in cpu_leaf.items()}
    rast = GaussianRasterizer(raster_settings=Hh.gpu_settings(scene, gpu))

    def render(frame_id, **kw):
        return rast(phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"], **kw)
    hip = FrameStep(net, leaf, mask.to(gpu), render, [u.to(gpu) for u in upstream], dist=None, num_frames=8)
    ref = FrameStep(cpu_net, cpu_leaf, mask, cpu_render, upstream, dist=None, num_frames=8, assemble=cpu_assemble)
    for frame in (2, 5):
        hip.zero_grad(); ref.zero_grad()
        o_hip, o_ref = hip(frame), ref(frame)
        assert hip.frame_time(frame) == frame / 7.0
        for i in (0, 1, 2, 4, 6):
            l1 = float((o_hip[i].detach().cpu().double() - o_ref[i].double()).abs().mean())
            assert l1 < 1e-5 * max(1.0, float(o_ref[i].abs().max())), (i, l1)
        assert torch.equal(o_hip[10].cpu(), o_ref[10])
        for k in leaf:
            assert _rel(leaf[k].grad.cpu().numpy(), cpu_leaf[k].grad.numpy()) < 1e-3, k
        assert _rel(hip.last[1].grad.cpu().numpy(), ref.last[1].grad.numpy()) < 1e-3
        for (name, p), (_, q) in zip(net.named_parameters(), cpu_net.named_parameters()):
            if q.grad is None:
                assert p.grad is None, name
            else:
                assert _rel(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-3, name
    assert hip.exchanges == 0


_RCCL_CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from gftorf_amd import reference_network
from gftorf_amd.deform import allreduce_gradients, flat_grad_bucket
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1, device_id=dev)
torch.manual_seed(1)
net = reference_network().to(dev)
n = 4096
x, t = torch.rand((n, 3), device=dev), torch.full((n, 1), 0.3, device=dev)
d_xyz, _, d_sh, _ = net(x, t)
torch.autograd.backward([d_xyz, d_sh], [torch.randn_like(d_xyz), torch.randn_like(d_sh)])
before = flat_grad_bucket(net)[0].clone()
nbytes = allreduce_gradients(net, dist, average=True)
after = flat_grad_bucket(net)[0]
torch.cuda.synchronize()
maps = open("/proc/self/maps").read()
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "bytes": nbytes,
                  "same": bool(torch.equal(before, after)), "nonzero": bool(before.abs().max() > 0),
                  "librccl_mapped": "librccl" in maps}))
dist.destroy_process_group()
'''


def test_single_rank_rccl_group_runs_the_exchange(gpu, tmp_path):
    """The path's one collective through RCCL itself: a fresh child process builds a world_size-1 `nccl` process group on
    cuda:0, runs the deformation network forward + backward on the device and `allreduce_gradients` through the group --
    librccl is loaded, a communicator is created, the all-reduce is launched on the gradient bucket (with one rank the
    averaged bucket equals the rank's own)."""
    import json
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    child = tmp_path / "rccl_child.py"
    child.write_text(_RCCL_CHILD % ROOT)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(child), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["backend"] == "nccl" and res["world"] == 1
    # the backward's gradients ARE the bucket (views of one buffer, each on a 16-byte boundary): the 516 915 values of the
    # used parameters plus one padding float behind xyz_warp.bias (3 values)
    assert res["bytes"] == (522055 - 5140 + 1) * 4
    assert res["nonzero"] and res["same"] and res["librccl_mapped"]


@pytest.mark.gpu
def test_bench_c4_runs_with_two_ranks(tmp_path):
    """`bench.py --workload C4 --gpus 2` as the driver launches it (torch.distributed.run), rehearsed with both ranks on
    this one GPU over gloo (GFT_BENCH_REHEARSAL=1): every composed step holds a collective, so all ranks must run the same
    number of steps in every leg -- a time-based spin-up once made the ranks' counts differ and the run hung."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.update(GFT_BENCH_REHEARSAL="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--workload", "C4", "--no-cpu-baseline", "--spin-up", "0.05"],
                       env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0
    assert line["deform_exchange"]["in_the_timed_step"] and line["deform_exchange"]["collectives_per_step"] == 1.0
    assert line["deform_exchange"]["ranks"] == 2 and line["roofline"]["kernel"].startswith("k_deform_fwd")
