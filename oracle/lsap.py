"""Oracle binding for oracle/lsap.c (test infrastructure).  `linear_sum_assignment(cost)` mirrors
scipy.optimize.linear_sum_assignment (minimisation) as the reference calls it
(/root/reference/model/box_utils.py:91, /root/reference/model/loss.py:92)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblsap_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "lsap.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.hh_oracle_lsap.restype = ctypes.c_int
        _lib.hh_oracle_lsap.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def linear_sum_assignment(cost):
    c = np.ascontiguousarray(np.asarray(cost, dtype=np.float64))
    if c.ndim != 2:
        raise ValueError("expected a matrix")
    nr, nc = c.shape
    k = min(nr, nc)
    rows = np.zeros(max(k, 1), dtype=np.int64)
    cols = np.zeros(max(k, 1), dtype=np.int64)
    n = _load().hh_oracle_lsap(nr, nc, c.ctypes.data, rows.ctypes.data, cols.ctypes.data)
    if n < 0:
        raise ValueError("cost matrix is infeasible")
    return rows[:n].copy(), cols[:n].copy()
