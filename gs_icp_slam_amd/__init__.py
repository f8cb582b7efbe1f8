"""gs_icp_slam_amd — MI355X (gfx950) implementation of GS-ICP-SLAM's per-frame hot path.

Sub-modules mirror the three native extension modules the reference imports:
  rasterizer -> diff_gaussian_rasterization   gicp -> pygicp   knn -> simple_knn._C
and are re-exported under those names by the drop-in packages at the repository root.  Everything computes in
libgsicp_hip.so (hand-written HIP, C ABI in include/gsicp_hip.h); there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
