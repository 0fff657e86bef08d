"""ORACLE (test infrastructure only) of simple_knn._C.distCUDA2 -- SURVEY section 8(f) row 3.

The reference (submodules/simple-knn/simple_knn.cu:148-183) finds, for every point, the three
smallest squared distances to the other points (exact: box pruning only skips boxes that cannot
improve the current third-best, :171-173) and returns their mean (:182).  Restated two ways:

* :func:`mean_dist2_bruteforce`: all pairs, float32 arithmetic in the reference's expression
  ``d.x*d.x + d.y*d.y + d.z*d.z`` (small P only);
* :func:`mean_dist2_kdtree`: scipy cKDTree for the neighbour indices (float64), distances
  recomputed in float32 as above (any P).

Only tests/ and bench.py's baseline leg may import this module.
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def _dist2_f32(a, b):
    d = (b.astype(np.float32) - a.astype(np.float32)).astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).astype(np.float32)


def mean_dist2_bruteforce(points):
    p = np.asarray(points, np.float32)
    P = p.shape[0]
    out = np.empty(P, np.float32)
    for i in range(P):
        d = _dist2_f32(p[i][None, :], p)
        d = np.delete(d, i)
        best = np.sort(d)[:3]
        best = np.concatenate([best, np.full(3 - best.size, FLT_MAX, np.float32)])
        with np.errstate(over="ignore"):
            out[i] = (best[0] + best[1] + best[2]) / np.float32(3.0)
    return out


def mean_dist2_kdtree(points):
    from scipy.spatial import cKDTree
    p = np.asarray(points, np.float32)
    P = p.shape[0]
    if P < 4:
        return mean_dist2_bruteforce(p)
    # k = 8 so that exact duplicates of the query (distance 0, possibly listed before the query
    # itself) never push a true neighbour out of the candidate set
    k = min(P, 8)
    _, idx = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
    out = np.empty(P, np.float32)
    rows = np.arange(P)[:, None]
    d = _dist2_f32(p[:, None, :], p[idx])
    d[idx == rows] = np.inf                                    # self excluded by index
    d.sort(axis=1)
    best = d[:, :3].astype(np.float32)
    return ((best[:, 0] + best[:, 1] + best[:, 2]) / np.float32(3.0)).astype(np.float32)
