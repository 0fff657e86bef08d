"""``distCUDA2`` of the reference's ``simple_knn`` extension on MI355X.

``scene/gaussian_model.py:194-199`` initialises the Gaussian scales from
``distCUDA2(points)`` = mean squared distance of every point to its three nearest neighbours
(``submodules/simple-knn/spatial.cu:14-25``).  Same name, argument and result here; the work is
done by ``csrc/k_knn.hip`` through ``include/gftorf_knn.h``.  There is no CPU path.
"""
import torch

from . import _lib


def distCUDA2(points):
    lib = _lib.load()
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    dev = points.device
    if dev.type != "cuda":
        raise RuntimeError("gftorf_amd.distCUDA2 runs on a HIP device only (points is on %s); there is no CPU path" % (dev,))
    pts = points.contiguous()
    if pts.dtype != torch.float32:
        pts = pts.float()
    P = pts.size(0)
    out = torch.empty((P,), device=dev, dtype=torch.float32)
    if P == 0:
        return out
    scratch = torch.empty((lib.gft_knn_scratch_bytes(P),), device=dev, dtype=torch.uint8)
    with _lib.on_device(dev):
        _lib.check(lib.gft_knn_mean_dist2(_lib.raw_stream(dev), P, pts.data_ptr(),
                                          out.data_ptr(), scratch.data_ptr()))
    return out
