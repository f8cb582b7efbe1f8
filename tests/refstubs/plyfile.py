"""Stand-in for plyfile (imported by scene/dataset_readers.py and scene/gaussian_model.py; used only by load_ply / save_ply,
which the SLAM loop reaches only with --save_results)."""


class _Unavailable:
    def __init__(self, *a, **k):
        raise RuntimeError("plyfile is not installed in this image (stand-in)")

    @staticmethod
    def describe(*a, **k):
        raise RuntimeError("plyfile is not installed in this image (stand-in)")

    @staticmethod
    def read(*a, **k):
        raise RuntimeError("plyfile is not installed in this image (stand-in)")


PlyData = _Unavailable
PlyElement = _Unavailable
