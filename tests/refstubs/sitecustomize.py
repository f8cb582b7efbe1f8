"""Environment shim (auto-imported by CPython's `site` when tests/refstubs is on PYTHONPATH) for the reference's UNTOUCHED host
files, which were written against NumPy 1.x ("resolve by environment rather than by patching them"):
  * `np.unicode_` (removed in NumPy 2)                                           [REF utils/traj_utils.py:129]
  * `np.linalg.inv(torch.Tensor)` returned an ndarray under NumPy 1 (the result was wrapped with ndarray.__array_wrap__); NumPy 2
    wraps it with the tensor's own __array_wrap__, and the next line calls ndarray-style `.transpose()` on it  [REF mp_Mapper.py:367-369]
Nothing here touches the hot path."""
import numpy as _np

if not hasattr(_np, "unicode_"):
    _np.unicode_ = _np.str_

_inv = _np.linalg.inv


def _inv_numpy1(a):
    if type(a).__module__.startswith("torch"):
        a = a.detach().cpu().numpy()
    return _inv(a)


_np.linalg.inv = _inv_numpy1
