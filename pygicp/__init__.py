"""Drop-in for the reference's ``pygicp`` extension module [REF mp_Tracker.py:10, 53]."""
from gs_icp_slam_amd.gicp import FastGICP  # noqa: F401
