"""Pins the oracle against every known-answer the reference offers for this path:
the SH colour polynomial and camera conventions of its importable Python helpers
(fixtures captured by tests/golden/make_golden.py) and the values recorded in
SURVEY.md Appendix B.  The rasterizer proper has no reference test."""
import math
import os

import numpy as np
import pytest

from gftorf_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _render_one_sh(oracle, sh, direction, deg, sh_p=None):
    """Run the oracle's K1 on one Gaussian placed along `direction` from the camera and
    return rgb (after +0.5 and clamp) and the clamp flags; with `sh_p` ([16, 2] phase / amplitude coefficients) the
    (phase_sh, amplitude) pair and the amplitude's clamp flag instead."""
    campos = np.zeros(3, np.float32)
    p = (direction / np.linalg.norm(direction) * 3.0).astype(np.float32)
    # camera looking along +z with the point in front: rotate world so that p maps to +z
    z = p / np.linalg.norm(p)
    x = np.cross([0.0, 1.0, 0.0], z)
    if np.linalg.norm(x) < 1e-6:
        x = np.cross([1.0, 0.0, 0.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = np.stack([x, y, z], 0)
    cam = synth.make_camera(64, 64, w2c=w2c)
    cfg = oracle.make_config(1, deg, 16, 0 if sh_p is None else 16, 64, 64, cam["tanfovx"], cam["tanfovy"], near_n=0.45,
                             far_n=6.05, use_view_dependent_phase=sh_p is not None)
    g = oracle.preprocess_fwd(cfg, p[None].copy(), np.full((1, 3), 0.05, np.float32),
                              np.array([[1, 0, 0, 0]], np.float32), np.array([0.5], np.float32),
                              np.ascontiguousarray(sh[None]), None if sh_p is None else np.ascontiguousarray(sh_p[None]),
                              None, None, None,
                              cam["viewmatrix"].reshape(-1), cam["projmatrix"].reshape(-1), campos)
    assert g["radii"][0] > 0
    if sh_p is not None:
        return g["phase_amp"][0].copy(), bool(g["clamped_p"][0])
    return g["rgb"][0], g["clamped"][0]


def test_sh_colour_matches_reference_eval_sh(oracle):
    d = np.load(os.path.join(GOLD, "sh_color.npz"))
    sh, dirs = d["sh"], d["dirs"]
    for deg in range(4):
        ref = d["deg%d" % deg]              # eval_sh output, before +0.5 / clamp
        for i in range(0, sh.shape[0], 4):
            rgb, clamped = _render_one_sh(oracle, sh[i], dirs[i], deg)
            want = ref[i] + 0.5
            np.testing.assert_allclose(rgb, np.maximum(want, 0.0), rtol=0, atol=3e-6)
            ok = np.abs(want) > 1e-5
            np.testing.assert_array_equal(clamped[ok].astype(bool), (want < 0)[ok])


def test_sh_phasor_matches_reference_eval_sh(oracle):
    """computePhasorFromSH (forward.cu:73-125) is the helper's polynomial on two channels: `+0.5` on both, then the
    phase loses `0.5 + C0 sh_p[0].x` again (only the view-dependent residual is left) and the amplitude is clamped at 0
    with its flag (forward.cu:113-124).  Pinned against utils/sh_utils.py:57-112 eval_sh, degrees 0-3."""
    d = np.load(os.path.join(GOLD, "sh_phasor.npz"))
    sh_p, dirs = d["sh_p"], d["dirs"]
    sh = np.zeros((16, 3), np.float32)
    seen = set()
    for deg in range(4):
        ref = d["deg%d" % deg]              # eval_sh output [n, 2], before +0.5
        for i in range(0, sh_p.shape[0], 2):
            (phase, amp), clamped = _render_one_sh(oracle, sh, dirs[i], deg, sh_p=sh_p[i])
            want_phase = ref[i, 0] - synth.SH_C0 * sh_p[i, 0, 0]
            want_amp = ref[i, 1] + 0.5
            np.testing.assert_allclose(phase, want_phase, rtol=0, atol=3e-6)
            np.testing.assert_allclose(amp, max(want_amp, 0.0), rtol=0, atol=3e-6)
            if abs(want_amp) > 1e-5:
                assert clamped == (want_amp < 0)
                seen.add(bool(clamped))
        if deg == 0:
            # degree 0: the phase residual is exactly the rounding of (C0 x + 0.5) - 0.5 - C0 x
            assert np.abs(ref[::2, 0] - synth.SH_C0 * sh_p[::2, 0, 0]).max() < 1e-6
    assert seen == {False, True}            # both sides of the amplitude clamp were met


def test_survey_known_answers(oracle):
    # coefficients sh[k][c] = (k+1)(c+1)/48 - 0.5, dir = normalize(1,2,3)
    k = np.arange(16)[:, None] + 1
    c = np.arange(3)[None, :] + 1
    sh = (k * c / 48.0 - 0.5).astype(np.float32)
    want = {0: (-0.13517043, -0.12929346, -0.12341648), 1: (-0.13244992, -0.12385243, -0.11525495),
            2: (-0.00774528, -0.05967784, -0.11161041), 3: (0.09881671, -0.10560812, -0.31003296)}
    for deg, w in want.items():
        rgb, _ = _render_one_sh(oracle, sh, np.array([1.0, 2.0, 3.0]), deg)
        np.testing.assert_allclose(rgb, np.maximum(np.array(w) + 0.5, 0), rtol=0, atol=3e-6)
    for n, b in [(256, 9), (300, 9), (1200, 11), (8160, 13), (1, 1), (2, 2), (65536, 17)]:
        assert oracle.get_higher_msb(n) == b


def test_camera_convention_matches_reference_helpers():
    d = np.load(os.path.join(GOLD, "camera.npz"))
    for name in ["bench640", "bench1080", "c1_256"]:
        W, H, fovx, fovy, zn, zf = d[name + "_args"]
        np.testing.assert_allclose(synth.projection_matrix(zn, zf, fovx, fovy), d[name + "_proj"], rtol=1e-6, atol=1e-7)
    # benchmark camera: full projection handed to the rasterizer
    cam = synth.make_camera(640, 480)
    assert abs(cam["tanfovx"] - 0.5773502691896257) < 1e-12 and abs(cam["tanfovy"] - 0.4330127018922193) < 1e-12
    want = [1.732050776, 0, 0, 0, 0, 2.309401035, 0, 0, 0, 0, 1.080357194, 1, 0, 0, -0.486160725, 0]
    np.testing.assert_allclose(cam["projmatrix"].reshape(-1), want, rtol=1e-6, atol=1e-7)
    # world-to-view of the reference (getWorld2View2) is [R^T | t]: our w2c is used as given
    R, t = d["w2v_R"], d["w2v_t"]
    M = np.eye(4)
    M[:3, :3] = R.T
    M[:3, 3] = t
    np.testing.assert_allclose(d["w2v"], M, atol=1e-6)
    c2 = synth.make_camera(64, 48, w2c=d["w2v"])
    np.testing.assert_allclose(c2["viewmatrix"], d["w2v"].T, atol=0)
    np.testing.assert_allclose(c2["campos"], np.linalg.inv(M)[:3, 3], atol=1e-6)


def test_pa2sh_rgb2sh_recipe():
    d = np.load(os.path.join(GOLD, "sh_color.npz"))
    x = d["rgb2sh_in"]
    np.testing.assert_allclose((x - 0.5) / synth.SH_C0, d["rgb2sh_out"], rtol=1e-6)
    np.testing.assert_allclose((x - 0.5) / synth.SH_C0, d["pa2sh_out"], rtol=1e-6)
