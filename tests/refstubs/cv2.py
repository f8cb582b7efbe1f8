"""Stand-in for OpenCV: the three calls on the reference's live path, on top of PIL + numpy.
  cv2.imread(path)                              [REF mp_Tracker.py:350, 356; gs_icp_slam.py:143, 150; mp_Mapper.py:361]
  cv2.imread(path, cv2.IMREAD_UNCHANGED)        [REF mp_Mapper.py:362]  (16-bit depth PNG, returned as stored)
  cv2.cvtColor(img, cv2.COLOR_RGB2BGR)          [REF mp_Tracker.py:120; mp_Mapper.py:364]  (channel reversal)
"""
import numpy as np
from PIL import Image

IMREAD_UNCHANGED = -1
IMREAD_COLOR = 1
COLOR_RGB2BGR = 4
COLOR_BGR2RGB = 4
INTER_NEAREST = 0


def imread(path, flags=IMREAD_COLOR):
    try:
        im = Image.open(path)
    except (FileNotFoundError, OSError):
        return None
    if flags == IMREAD_UNCHANGED:
        a = np.array(im)
        return a[..., ::-1].copy() if a.ndim == 3 else a
    return np.array(im.convert("RGB"))[..., ::-1].copy()   # OpenCV hands out B,G,R


def cvtColor(img, code):
    if code != COLOR_RGB2BGR:
        raise NotImplementedError("cv2 stand-in: only COLOR_RGB2BGR / COLOR_BGR2RGB")
    return np.ascontiguousarray(img[..., ::-1])


def resize(img, size, interpolation=None):
    w, h = size
    ys = (np.arange(h) * img.shape[0] / h).astype(np.int64)
    xs = (np.arange(w) * img.shape[1] / w).astype(np.int64)
    return img[ys][:, xs]
