"""Drop-in import name used by the reference's callers
(``gaussian_renderer/__init__.py:14``:
``from diff_gaussian_rasterization_w_tof import GaussianRasterizationSettings, GaussianRasterizer``).
Everything is implemented in :mod:`gftorf_amd`."""
from gftorf_amd.api import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                            rasterize_gaussians, _RasterizeGaussians, cpu_deep_copy_tuple)
