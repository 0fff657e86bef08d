"""Seeded synthetic ToF-Gaussian scenes (numpy only).

The reference ships no datasets that are reachable here (no network), so
tests and bench.py render synthetic Gaussians of the shapes BASELINE.json names.
Camera conventions follow the reference: matrices are stored transposed
(``scene/cameras.py:121-129``) and the projection is
``utils/graphics_utils.py:55-75`` (restated, pinned by tests/golden/).
Recipe: SURVEY.md section 8(d).
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:55-75 restated (math convention, row-major)."""
    t = math.tan(fovy / 2) * znear
    r = math.tan(fovx / 2) * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * r)
    P[1, 1] = 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_w2c(yaw=0.0, pitch=0.0, roll=0.0, t=(0.0, 0.0, 0.0)):
    """World-to-camera 4x4 (math convention) from Euler angles + translation."""
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cr, sr = math.cos(roll), math.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    R = Rz @ Rx @ Ry
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = np.asarray(t, np.float64)
    return M.astype(np.float32)


def make_camera(W, H, fovx_deg=60.0, znear=0.45, zfar=6.05, w2c=None):
    """Returns dict(viewmatrix, projmatrix, campos, tanfovx, tanfovy, ...) with the
    4x4s in the reference's transposed storage."""
    tanfovx = math.tan(math.radians(fovx_deg) * 0.5)
    tanfovy = tanfovx * H / W
    fovx = 2 * math.atan(tanfovx)
    fovy = 2 * math.atan(tanfovy)
    if w2c is None:
        w2c = np.eye(4, dtype=np.float32)
    Pm = projection_matrix(znear, zfar, fovx, fovy)
    view_t = np.ascontiguousarray(w2c.T.astype(np.float32))           # world_view_transform
    full_t = np.ascontiguousarray((view_t @ Pm.T).astype(np.float32))  # full_proj_transform
    campos = np.linalg.inv(w2c.astype(np.float64))[:3, 3].astype(np.float32)
    return dict(viewmatrix=view_t, projmatrix=full_t, campos=campos,
                tanfovx=tanfovx, tanfovy=tanfovy, znear=znear, zfar=zfar,
                image_width=W, image_height=H, w2c=w2c)


def make_gaussians(P, cam, seed, sh_coeffs=16, scale_lo=0.002, scale_hi=0.02,
                   z_lo=1.0, z_hi=5.5, spread=1.05, cluster=0.0, opacity_range=None):
    """Gaussians placed in the camera frustum (camera space), mapped to world by
    the inverse of cam['w2c'].  Returns a dict of float32 arrays."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(z_lo, z_hi, P)
    u = rng.uniform(-spread, spread, P)
    v = rng.uniform(-spread, spread, P)
    if cluster > 0.0:
        # screen-space density falling off from the image centre (robustness workloads only;
        # the BASELINE configs are uniform)
        u = np.clip(rng.normal(0.0, cluster, P), -spread, spread)
        v = np.clip(rng.normal(0.0, cluster, P), -spread, spread)
    pc = np.stack([u * z * cam["tanfovx"], v * z * cam["tanfovy"], z, np.ones(P)], 1)
    c2w = np.linalg.inv(cam["w2c"].astype(np.float64))
    means3D = (pc @ c2w.T)[:, :3].astype(np.float32)
    scales = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4))
    rotations = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacities = (1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32)
    if opacity_range is not None:
        # thin scenes: the reference initialises every opacity to 0.1 (arguments/__init__.py:99,
        # scene/gaussian_model.py:202) -- nothing saturates, every list is walked whole.  Drawn from a
        # generator of its own so that every other attribute equals the default scene's.
        lo, hi = opacity_range
        opacities = np.random.default_rng(seed + 31337).uniform(lo, hi, (P, 1)).astype(np.float32)
    M = sh_coeffs
    shs = np.zeros((P, M, 3), np.float32)
    shs[:, 0, :] = rng.uniform(-1, 1, (P, 3)) / SH_C0 * 0.5
    if M > 1:
        shs[:, 1:, :] = rng.normal(0, 0.1, (P, M - 1, 3))
    shs_p = np.zeros((P, M, 2), np.float32)
    shs_p[:, 0, 0] = (rng.uniform(0, 2 * math.pi, P) - 0.5) / SH_C0   # PA2SH(phase)
    shs_p[:, 0, 1] = (rng.uniform(0.05, 0.5, P) - 0.5) / SH_C0       # PA2SH(amplitude)
    if M > 1:
        shs_p[:, 1:, :] = rng.normal(0, 0.05, (P, M - 1, 2))
    return dict(means3D=means3D, scales=scales, rotations=rotations,
                opacities=opacities, shs=shs, shs_p=shs_p)


def make_background(W, H, seed):
    rng = np.random.default_rng(seed + 7919)
    return (rng.random((7, H, W), dtype=np.float32) * 2 - 1).astype(np.float32)


def make_pixel_grads(W, H, seed):
    """Fixed upstream gradients that exercise every differentiable output."""
    rng = np.random.default_rng(seed + 104729)
    g = lambda c: rng.normal(0, 1, (c, H, W)).astype(np.float32)
    return dict(color=g(3), phasor=g(7), depth=g(1), acc=g(1), depth_distortion=g(1))


# BASELINE.json configs (SURVEY.md 8(d) config map)
CONFIGS = {
    "C1": dict(P=10_000, W=256, H=256, D=0, sh_coeffs=1, tof=False),
    "C2": dict(P=500_000, W=640, H=480, D=3, sh_coeffs=16, tof=True),
    "metric": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True),
    "C5": dict(P=5_000_000, W=1920, H=1080, D=3, sh_coeffs=16, tof=True),
    # the metric frame with the opacities the reference's scenes start from: nothing saturates
    "fog": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True, opacity_range=(0.05, 0.1)),
}


def make_scene(name_or_cfg, seed=1234, w2c=None, P=None):
    cfg = dict(CONFIGS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)
    if P is not None:
        cfg["P"] = P
    cam = make_camera(cfg["W"], cfg["H"], w2c=w2c)
    g = make_gaussians(cfg["P"], cam, seed, sh_coeffs=cfg["sh_coeffs"], cluster=cfg.get("cluster", 0.0),
                       opacity_range=cfg.get("opacity_range"))
    if not cfg.get("tof", True):
        g["shs_p"] = None
    return dict(cfg=cfg, cam=cam, gaussians=g, bg=make_background(cfg["W"], cfg["H"], seed),
                grads=make_pixel_grads(cfg["W"], cfg["H"], seed),
                depth_range=10.0, phase_offset=0.1, dc_offset=0.05,
                use_view_dependent_phase=True)


# ---- deformation network: seeded parameters (synthetic data, like the scenes above) ----------------------------
def deform_param_shapes(t_multires=10, xyz_multires=10, D=8, W=256, num_shs=16):
    """``state_dict`` names -> shapes of the reference's ``DeformNetwork`` (utils/time_utils.py:68-81): D hidden layers of
    width W, the encoded input re-injected behind layer D // 2, heads xyz_warp 3, rot 4, r / g / b / a num_shs."""
    n_in = 3 + 6 * xyz_multires + 1 + 2 * t_multires
    s = {}
    for i in range(D):
        fan_in = n_in if i == 0 else (W + n_in if i == D // 2 + 1 else W)
        s["linear.%d.weight" % i] = (W, fan_in)
        s["linear.%d.bias" % i] = (W,)
    for name, out in (("xyz_warp", 3), ("rot", 4), ("r", num_shs), ("g", num_shs), ("b", num_shs), ("a", num_shs)):
        s[name + ".weight"] = (out, W)
        s[name + ".bias"] = (out,)
    return s


def random_deform_params(seed, head_std=0.05, t_multires=10):
    """Seeded parameters of a usable magnitude (Xavier-like trunk; heads larger than the reference's 1e-5 initialisation so
    that outputs and gradients are well away from zero): numpy arrays under the reference's ``state_dict`` names, torch's
    ``[out, in]`` layout."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in deform_param_shapes(t_multires).items():
        if name.endswith(".bias"):
            p[name] = rng.normal(0.0, 0.02, shape).astype(np.float32)
        elif name.startswith("linear."):
            p[name] = rng.normal(0.0, np.sqrt(2.0 / (shape[0] + shape[1])), shape).astype(np.float32)
        else:
            p[name] = rng.normal(0.0, head_std, shape).astype(np.float32)
    return p
