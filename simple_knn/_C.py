"""Drop-in for ``simple_knn._C`` [REF scene/gaussian_model.py:20]."""
from gs_icp_slam_amd.knn import distCUDA2  # noqa: F401
