"""Stand-in for open3d: `np.array(o3d.io.read_image(path))` [REF mp_Tracker.py:351, 357; gs_icp_slam.py:144, 151]."""
from . import io  # noqa: F401
