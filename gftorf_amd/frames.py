"""One data-parallel iteration of the multi-frame (ftorf) configuration: BASELINE.json config 4, SURVEY section 8(e).

Frames are the independent units of the path: every rank holds a replica of the Gaussians and of the deformation
network and renders its own frame.  What one rank does per iteration is the reference's dynamic branch
(``train.py:164-178`` -> ``gaussian_renderer/__init__.py:81-128``):

    d_xyz, d_rot, d_sh, d_sh_p = query_dmlp(frame_id / (total_num_views - 1))      # the network at the frame's time
    means3D, ..., shs_p = static rows | dynamic rows + offsets                      # input assembly
    outputs = rasterizer(...)                                                       # forward
    loss.backward()                                                                 # rasterizer -> assembly -> network

and then the ONE exchange of the path: an all-reduce of the network's gradient bucket (RCCL over xGMI under the
``nccl`` backend), after which every replica holds the same gradients.  The rasterizer itself has no collective.

:class:`FrameStep` composes exactly that from this package's pieces; the pieces are arguments so that the same step
can be rehearsed on CPU ranks (gloo) with host-side stand-ins (``tests/test_dist_gloo.py``) -- the product default
is the HIP path and nothing else.
"""
import torch

from .assemble import assemble_inputs
from .deform import allreduce_gradients

# outputs of the rasterizer that carry a gradient: color, phasor, depth, acc, depth_distortion (11-tuple positions)
DIFFERENTIABLE_OUTPUTS = (0, 1, 2, 4, 6)


class FrameStep:
    """``step = FrameStep(net, gaussians, motion_mask, render, upstream, dist)``; ``step(frame_id)`` runs one
    iteration for that frame and leaves the gradients on the Gaussian leaves and (all-reduced) on the network.

    net         the deformation network (``gftorf_amd.DeformNetwork``); ``net(x, t) -> d_xyz, d_rot, d_sh, d_sh_p``
    gaussians   dict of leaf tensors ``xyz, opacity, scaling, rotation_raw, fc, fp`` (activated values, as
                ``GaussianModel.get_*`` returns them; ``rotation_raw`` is normalised here)
    motion_mask bool[P]: the dynamic Gaussians (``pc.get_motion_mask``)
    render      ``render(frame_id, means3D=, means2D=, opacities=, shs=, shs_p=, scales=, rotations=) -> 11-tuple``
                (a ``GaussianRasterizer`` of the frame's camera)
    upstream    the five upstream gradients of ``DIFFERENTIABLE_OUTPUTS`` (a fixed stand-in for the loss)
    dist        ``torch.distributed`` (initialised) or None: no exchange (single replica)
    """

    def __init__(self, net, gaussians, motion_mask, render, upstream, dist=None, num_frames=8,
                 assemble=assemble_inputs, exchange=allreduce_gradients):
        self.net, self.g, self.mask, self.render, self.upstream = net, gaussians, motion_mask, render, list(upstream)
        self.dist, self.num_frames, self.assemble, self.exchange = dist, int(num_frames), assemble, exchange
        xyz = gaussians["xyz"].detach()
        lo, hi = xyz.min(0).values, xyz.max(0).values
        # get_xyz_normalized of the dynamic rows, detached (scene/gaussian_model.py:170-174)
        self.x_norm = ((xyz - lo) / (hi - lo))[motion_mask].contiguous()
        self.exchanges = 0          # collectives issued so far: one per iteration
        self.exchanged_bytes = 0
        self.last = None
        from .deform import DeformNetwork
        self._scalar_zeros = isinstance(net, DeformNetwork) and assemble is assemble_inputs
        self._ssp = None
        self.mark = None            # optional `mark(name)` called at the phase boundaries (bench.py records events there)

    def frame_time(self, frame_id):
        return float(frame_id % self.num_frames) / max(self.num_frames - 1, 1)

    def __call__(self, frame_id):
        g, dev = self.g, self.g["xyz"].device
        n = self.x_norm.size(0)
        mark = self.mark or (lambda name: None)
        mark("start")
        t = torch.full((1, 1), self.frame_time(frame_id), device=dev, dtype=torch.float32).expand(n, -1)
        # (this package's network hands the two all-zero offsets over as the scalar 0.0, as train.py:164 does for a static scene)
        d_xyz, d_rot, d_sh, d_sh_p = self.net(self.x_norm, t, zeros_as_scalars=True) if self._scalar_zeros else self.net(self.x_norm, t)
        mark("network_forward")
        # screenspace_points (gaussian_renderer/__init__.py:52-56): zeros whose only role is to receive a gradient -- the same
        # leaf every step (its values are never written), the gradient of the last step dropped
        ssp = self._ssp
        if ssp is None or ssp.size(0) != g["xyz"].size(0) or ssp.device != dev:
            ssp = self._ssp = torch.zeros((g["xyz"].size(0), 3), device=dev, dtype=torch.float32, requires_grad=True)
        ssp.grad = None
        # pc.get_rotation: this package's assembly normalises the static rows itself (rotation=None); a stand-in gets the
        # activated tensor as the reference's renderer does
        rot = None if self.assemble is assemble_inputs else torch.nn.functional.normalize(g["rotation_raw"])
        m3, m2, op, sc, ro, shs, shp = self.assemble(g["xyz"], ssp, g["opacity"], g["scaling"], rot, g["rotation_raw"],
                                                     g["fc"], g["fp"], self.mask, d_xyz, d_rot, d_sh, d_sh_p)
        mark("assembly")
        outs = self.render(frame_id, means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc, rotations=ro)
        mark("raster_forward")
        torch.autograd.backward([outs[i] for i in DIFFERENTIABLE_OUTPUTS], self.upstream)
        mark("backward")               # rasterizer -> assembly -> network
        if self.dist is not None:
            self.exchanged_bytes = self.exchange(self.net, self.dist, average=True)
            self.exchanges += 1
        mark("exchange")
        self.last = (outs, ssp)
        return outs

    def zero_grad(self):
        for v in self.g.values():
            v.grad = None
        self.net.zero_grad(set_to_none=True)
