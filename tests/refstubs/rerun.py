"""Stand-in for rerun-sdk: the reference only touches it behind `--rerun_viewer` [REF mp_Tracker.py:103-105; gs_icp_slam.py:48-50]."""


def __getattr__(name):
    def _noop(*a, **k):
        return None
    return _noop
