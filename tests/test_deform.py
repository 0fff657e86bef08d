"""Deformation network: oracle vs the reference's golden vectors (CPU),
HIP path vs oracle (GPU, through the C ABI in include/gftorf_deform.h)."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "deform.npz")


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


# ---------------------------------------------------------------------------------------------
# CPU: the oracle is pinned by outputs and autograd gradients of the reference's own module
# ---------------------------------------------------------------------------------------------
def test_oracle_matches_reference_module_forward():
    g = np.load(GOLDEN)
    params = deform_ref.random_params(int(g["seed"]))
    d_xyz, d_rot, d_sh, d_sh_p = deform_ref.forward(params, g["x"], g["t"])
    assert d_xyz.shape == g["d_xyz"].shape and d_sh.shape == g["d_sh"].shape == (g["x"].shape[0], 16, 3)
    assert _rel(d_xyz, g["d_xyz"]) < 2e-6
    assert _rel(d_sh, g["d_sh"]) < 2e-6
    # the reference returns zeros for the rotation and phasor offsets (time_utils.py:127)
    assert d_rot.shape == g["d_rot"].shape and not d_rot.any() and not g["d_rot"].any()
    assert d_sh_p.shape == g["d_sh_p"].shape and not d_sh_p.any() and not g["d_sh_p"].any()


def test_oracle_matches_reference_module_backward():
    g = np.load(GOLDEN)
    params = deform_ref.random_params(int(g["seed"]))
    grads = deform_ref.backward(params, g["x"], g["t"], g["g_dxyz"], g["g_dsh"])
    assert sorted(n for n, v in grads.items() if v is None) == sorted(g["grad_none"].tolist())
    seen = 0
    for key in g.files:
        if key.startswith("grad:"):
            assert _rel(grads[key[5:]], g[key]) < 5e-6, key
            seen += 1
        elif key.startswith("grad_s:"):
            assert _rel(grads[key[7:]][::8, ::4], g[key]) < 5e-6, key
            seen += 1
    assert seen == 2 * 8 + 2 * 4


def test_oracle_float32_close_to_float64():
    params = deform_ref.random_params(5)
    rng = np.random.default_rng(6)
    x, t = rng.random((33, 3)).astype(np.float32), rng.random((33, 1)).astype(np.float32)
    a = deform_ref.forward(params, x, t)
    b = deform_ref.forward(params, x, t, dtype=np.float64)
    assert _rel(a[0], b[0]) < 5e-6 and _rel(a[2], b[2]) < 5e-6


def test_embedding_layout():
    x = np.array([[0.1, 0.2, 0.3]], np.float32)
    t = np.array([[0.5]], np.float32)
    e = deform_ref.embed(x, t)
    assert e.shape == (1, 76)
    np.testing.assert_array_equal(e[0, :3], x[0])
    np.testing.assert_array_equal(e[0, 3:6], np.sin(x[0]))            # frequency 1: sin of all dims, then cos
    np.testing.assert_array_equal(e[0, 6:9], np.cos(x[0]))
    np.testing.assert_array_equal(e[0, 57:60], np.sin(x[0] * np.float32(512)))
    assert e[0, 63] == t[0, 0]
    np.testing.assert_array_equal(e[0, 64:66], [np.sin(t[0, 0]), np.cos(t[0, 0])])
    np.testing.assert_array_equal(e[0, 74:76], [np.sin(t[0, 0] * np.float32(32)), np.cos(t[0, 0] * np.float32(32))])
