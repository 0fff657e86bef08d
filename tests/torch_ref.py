"""Dense float64 PyTorch re-derivation of the ToF Gaussian rasterizer (test helper).

Independent of both the oracle (C) and the HIP kernels: the forward equations of
SURVEY.md Appendix A.2/A.3 written with differentiable torch ops in float64;
gradients come from autograd, not from the reference's hand-written backward.
Used to check that the oracle's backward (a restatement of
RAST/cuda_rasterizer/backward.cu) is the gradient of its forward.

Deliberate gradient conventions of the reference that autograd must mimic:
  * alpha = min(0.99, o*G): the clamp is ignored by the backward
    (backward.cu:752,869) -> straight-through.
  * t.x = clamp(t.x/t.z)*t.z: d/dt.x is gated by x_grad_mul, d/dt.z of the
    clamped value is dropped (backward.cu:293-297,383-385).
  * phase DC removal "- C0*sh_p[0].x" (forward.cu:115) is not differentiated
    (backward.cu:168-169).
Discrete decisions (culling, radii, tile lists, skips, termination) are taken
from the oracle's forward so both sides blend the same lists.
"""
import math

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
      -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
      0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435]


def sh_poly(deg, sh, dirs):
    """sh [P,M,C], dirs [P,3] -> [P,C] (kernel layout [coeff][channel])."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5]
               + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + C2[3] * xz * sh[:, 7]
               + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
               + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
               + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def render(params, fwd, settings):
    """params: dict of float64 leaf tensors (means3D, opacities, shs|colors_precomp,
    shs_p|phasors_precomp, scales+rotations|cov3D_precomp, phase_offset, dc_offset).
    fwd: oracle ForwardResult (lists / masks).  settings: dict of scalars + matrices.
    Returns dict of output images (float64) + 'ndc' (retain_grad leaf-like)."""
    dt = torch.float64
    W, H = settings["image_width"], settings["image_height"]
    V = torch.tensor(np.asarray(settings["viewmatrix"]).reshape(4, 4), dtype=dt)  # transposed storage
    PV = torch.tensor(np.asarray(settings["projmatrix"]).reshape(4, 4), dtype=dt)
    campos = torch.tensor(np.asarray(settings["campos"]).reshape(3), dtype=dt)
    tanfovx, tanfovy = settings["tanfovx"], settings["tanfovy"]
    focal_x, focal_y = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    near_n, far_n = settings["near_n"], settings["far_n"]
    D = settings["sh_degree"]
    mod = settings.get("scale_modifier", 1.0)
    dist2phase = 4.0 * math.pi / settings["depth_range"]
    vdp = settings["use_view_dependent_phase"]

    p = params["means3D"]
    P = p.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([p, ones], 1)
    p_view = (ph @ V)[:, :3]           # stored-transposed: row-vector times matrix
    p_hom = ph @ PV
    p_w = 1.0 / (p_hom[:, 3:4] + 1e-7)
    ndc = p_hom[:, :2] * p_w
    ndc.retain_grad()
    pix_x = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    pix_y = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    # cov3D
    if params.get("cov3D_precomp") is not None:
        c6 = params["cov3D_precomp"]
        Sigma = torch.stack([torch.stack([c6[:, 0], c6[:, 1], c6[:, 2]], -1),
                             torch.stack([c6[:, 1], c6[:, 3], c6[:, 4]], -1),
                             torch.stack([c6[:, 2], c6[:, 4], c6[:, 5]], -1)], -2)
    else:
        s = params["scales"] * mod
        q = params["rotations"]
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rm = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
        Sigma = Rm @ torch.diag_embed(s * s) @ Rm.transpose(1, 2)

    # cov2D (EWA)
    tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = tx / tz, ty / tz
    gx_mul = ((txtz >= -limx) & (txtz <= limx)).to(dt)
    gy_mul = ((tytz >= -limy) & (tytz <= limy)).to(dt)
    txc = torch.clamp(txtz, -limx, limx) * tz
    tyc = torch.clamp(tytz, -limy, limy) * tz
    txu = tx * gx_mul + (txc - tx * gx_mul).detach()
    tyu = ty * gy_mul + (tyc - ty * gy_mul).detach()
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([focal_x / tz, zero, -(focal_x * txu) / (tz * tz)], -1),
                     torch.stack([zero, focal_y / tz, -(focal_y * tyu) / (tz * tz)], -1)], -2)  # [P,2,3]
    Rv = V[:3, :3].T  # math W2C rotation
    A = J @ Rv
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    con_x, con_y, con_z = c / det, -b / det, a / det

    # colour
    dir_orig = p - campos
    dirs = dir_orig / dir_orig.norm(dim=1, keepdim=True)
    if params.get("shs") is not None:
        rgb = torch.clamp(sh_poly(D, params["shs"], dirs) + 0.5, min=0.0)
    else:
        rgb = params["colors_precomp"]

    dist = p_view.norm(dim=1)
    d_ndc = far_n / (far_n - near_n) * (1 - near_n / dist)
    factor = 1.0 / (dist * dist)
    phase_offset, dc_offset = params["phase_offset"], params["dc_offset"]
    if params.get("phasors_precomp") is not None:
        pp = params["phasors_precomp"]
        phase = dist * dist2phase
        if vdp:
            phase = phase + pp[:, 0]
        amp = pp[:, 1]
    elif params.get("shs_p") is not None:
        sp = params["shs_p"]
        pa = sh_poly(D, sp, dirs) + 0.5
        phase_sh = pa[:, 0] - 0.5 - (C0 * sp[:, 0, 0]).detach()
        amp = torch.clamp(pa[:, 1], min=0.0)
        phase = dist * dist2phase + phase_offset
        if vdp:
            phase = phase + phase_sh
    else:
        phase = None
    if phase is not None:
        cp, sn = torch.cos(phase), torch.sin(phase)
        ph7 = torch.stack([cp * amp * factor, sn * amp * factor, amp * factor,
                           (cp + dc_offset) * amp * factor, (-cp + dc_offset) * amp * factor,
                           (sn + dc_offset) * amp * factor, (-sn + dc_offset) * amp * factor], -1)
    else:
        ph7 = torch.zeros(P, 7, dtype=dt)
    opac = params["opacities"].reshape(-1)

    bg = torch.tensor(np.broadcast_to(np.asarray(settings["bg"], np.float64), (7, H, W)).copy(), dtype=dt)
    out_color = torch.zeros(3, H, W, dtype=dt)
    out_phasor = torch.zeros(7, H, W, dtype=dt)
    out_depth = torch.zeros(1, H, W, dtype=dt)
    out_acc = torch.zeros(1, H, W, dtype=dt)
    out_dd = torch.zeros(1, H, W, dtype=dt)
    colors_l, phasors_l, depth_l, acc_l, dd_l, idx_l = [], [], [], [], [], []

    gxn, gyn = (W + 15) // 16, (H + 15) // 16
    ranges = fwd.ranges
    plist = torch.tensor(fwd.point_list.astype(np.int64))
    for tile in range(gxn * gyn):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        tx0, ty0 = (tile % gxn) * 16, (tile // gxn) * 16
        ys, xs = torch.meshgrid(torch.arange(ty0, min(ty0 + 16, H)), torch.arange(tx0, min(tx0 + 16, W)), indexing="ij")
        xs, ys = xs.reshape(-1), ys.reshape(-1)
        n = xs.numel()
        pxf, pyf = xs.to(dt), ys.to(dt)
        T = torch.ones(n, dtype=dt)
        done = torch.zeros(n, dtype=torch.bool)
        Cc = torch.zeros(n, 3, dtype=dt)
        Pp = torch.zeros(n, 7, dtype=dt)
        Dd = torch.zeros(n, dtype=dt)
        Aa = torch.zeros(n, dtype=dt)
        DD = torch.zeros(n, dtype=dt)
        DD_D = torch.zeros(n, dtype=dt)
        DD_D2 = torch.zeros(n, dtype=dt)
        for k in range(r0, r1):
            if bool(done.all()):
                break
            j = int(plist[k])
            dx = pix_x[j] - pxf
            dy = pix_y[j] - pyf
            power = -0.5 * (con_x[j] * dx * dx + con_z[j] * dy * dy) - con_y[j] * dx * dy
            G = torch.exp(power)
            a_raw = opac[j] * G
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
            skip = (power > 0) | (alpha < 1.0 / 255.0)
            test_T = T * (1 - alpha)
            newly_done = (~skip) & (~done) & (test_T < 1e-4)
            done = done | newly_done
            act = (~skip) & (~done)
            am = act.to(dt)
            w = alpha * T * am
            w_p = alpha * T * T * am
            Cc = Cc + w[:, None] * rgb[j][None, :]
            Pp = Pp + w_p[:, None] * ph7[j][None, :]
            Dd = Dd + w * dist[j]
            zz = d_ndc[j]
            DD = DD + w * (zz * zz * Aa - 2 * zz * DD_D + DD_D2)
            DD_D = DD_D + w * zz
            DD_D2 = DD_D2 + w * zz * zz
            Aa = Aa + w
            T = torch.where(act, test_T, T)
        colors_l.append(Cc + T[:, None] * bg[:3, ys, xs].T)
        phasors_l.append(Pp + T[:, None] * bg[:7, ys, xs].T)
        depth_l.append(Dd)
        acc_l.append(Aa)
        dd_l.append(DD)
        idx_l.append((ys, xs))
    for (ys, xs), cc, pp, dd_, aa, ddd in zip(idx_l, colors_l, phasors_l, depth_l, acc_l, dd_l):
        out_color[:, ys, xs] = cc.T
        out_phasor[:, ys, xs] = pp.T
        out_depth[0, ys, xs] = dd_
        out_acc[0, ys, xs] = aa
        out_dd[0, ys, xs] = ddd
    return dict(color=out_color, phasor=out_phasor, depth=out_depth, acc=out_acc,
                depth_distortion=out_dd, ndc=ndc)
