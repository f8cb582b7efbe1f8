"""Drop-in for the reference's ``diff_gaussian_rasterization`` package [REF gaussian_renderer/__init__.py:14]."""
from gs_icp_slam_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                        rasterize_gaussians)
