"""simple_knn.distCUDA2: oracle pins on CPU, HIP parity on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import knn_ref as K   # noqa: E402


def cloud(P, seed=0, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    if kind == "clustered":          # SfM-like: dense blobs + sparse background, far from the origin
        c = rng.normal(0, 3, (8, 3))
        pts = c[rng.integers(0, 8, P)] + rng.normal(0, 0.05, (P, 3))
        pts[: P // 10] = rng.uniform(-10, 10, (P // 10, 3))
        return (pts + 20.0).astype(np.float32)
    if kind == "plane":              # one flat axis: the Morton scale of that axis degenerates
        p = rng.uniform(0, 1, (P, 3)).astype(np.float32)
        p[:, 2] = 0.5
        return p
    if kind == "duplicates":
        p = rng.uniform(-1, 1, (P // 4 + 1, 3)).astype(np.float32)
        return np.concatenate([p, p, p, p])[:P]
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "plane", "duplicates"])
def test_kdtree_restatement_equals_bruteforce(kind):
    p = cloud(700, 1, kind)
    a, b = K.mean_dist2_bruteforce(p), K.mean_dist2_kdtree(p)
    np.testing.assert_allclose(b, a, rtol=2e-6, atol=1e-12)


def test_known_answers():
    # four corners of a unit square: each point sees two neighbours at 1 and one at 2
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    np.testing.assert_allclose(K.mean_dist2_bruteforce(p), np.full(4, 4.0 / 3.0, np.float32), rtol=1e-7)
    # fewer than three other points leave FLT_MAX terms in the mean (simple_knn.cu:153,182)
    assert np.all(K.mean_dist2_bruteforce(p[:3]) > 1e37)


def test_drop_in_import_name_and_cpu_rejection():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="HIP device only"):
        distCUDA2(torch.zeros(10, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("P,kind", [(1, "uniform"), (3, "uniform"), (4, "uniform"), (1000, "uniform"), (1025, "plane"),
                                   (5000, "duplicates"), (50_000, "clustered"), (200_000, "uniform")])
def test_hip_knn_vs_oracle(P, kind, gpu):
    from simple_knn._C import distCUDA2
    p = cloud(P, 5, kind)
    got = distCUDA2(torch.tensor(p, device=gpu)).cpu().numpy()
    ref = K.mean_dist2_kdtree(p)
    assert got.shape == ref.shape and got.dtype == np.float32
    big = ref > 1e37
    assert np.array_equal(big, (got > 1e37) | ~np.isfinite(got))
    # squared distances are sums of three products: contraction order (fma) changes the last bit
    np.testing.assert_allclose(got[~big], ref[~big], rtol=2e-6, atol=1e-12)


@pytest.mark.gpu
def test_hip_knn_is_exact_and_deterministic(gpu):
    """The counting-sort placement is arbitrary inside a cell; results must not depend on it."""
    from gftorf_amd import distCUDA2
    p = torch.tensor(cloud(100_000, 9, "clustered"), device=gpu)
    a = distCUDA2(p)
    for _ in range(3):
        assert torch.equal(distCUDA2(p), a)
    assert distCUDA2(torch.zeros(0, 3, device=gpu)).shape == (0,)
