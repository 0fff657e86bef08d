"""gftorf_amd -- MI355X-native differentiable ToF Gaussian rasterizer.

Drop-in for the reference's ``diff_gaussian_rasterization_w_tof`` extension
(brownvc/gftorf, submodules/diff-gaussian-rasterization-w-tof): same Python
operator API, hand-written gfx950 HIP kernels behind a C ABI
(include/gftorf_rast.h).  See DESIGN.md.
"""
from .api import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                  _RasterizeGaussians)
from .pair import GaussianRasterizerPair, render_pair  # noqa: F401
from .assemble import assemble_inputs, assemble_parameters  # noqa: F401
from .knn import distCUDA2  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .deform import DeformNetwork, REFERENCE_ARCH, reference_network  # noqa: F401
from . import densify  # noqa: F401
from . import loss  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "GaussianRasterizerPair", "render_pair", "assemble_inputs", "assemble_parameters", "distCUDA2", "FusedAdam",
           "DeformNetwork", "REFERENCE_ARCH", "reference_network", "densify"]
