"""ORACLE (test infrastructure, NOT product code) -- ctypes front-end of the plain-C
restatement in oracle/hrnet_oracle.c (HRNet.forward hrnet.py:157-189 and the decode
loop SimpleHRNet.py:297-308).  Only tests/, smoke() and bench.py's cpu_baseline may
import this."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhrnet_oracle.so")


class _Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float))]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hrnet_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libhrnet_oracle.so"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.hrnet_oracle_forward.restype = ctypes.c_int
        lib.hrnet_oracle_forward.argtypes = [ctypes.POINTER(_Tensor), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
        lib.hrnet_oracle_decode.restype = None
        lib.hrnet_oracle_decode.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_float)]
        _lib = lib
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def hrnet_forward(sd: Dict[str, np.ndarray], images: np.ndarray, c: int, joints: int = 17,
                  acc_double: bool = True) -> np.ndarray:
    lib = _load()
    keep = []
    items = []
    for k, v in sd.items():
        a = np.asarray(v)
        if a.dtype != np.float32:
            continue  # num_batches_tracked
        a = np.ascontiguousarray(a)
        keep.append(a)
        items.append(_Tensor(k.encode(), _fp(a)))
    arr = (_Tensor * len(items))(*items)
    x = np.ascontiguousarray(images, dtype=np.float32)
    n, _, h, w = x.shape
    out = np.empty((n, joints, h // 4, w // 4), dtype=np.float32)
    rc = lib.hrnet_oracle_forward(arr, len(items), c, joints, _fp(x), n, h, w, _fp(out), int(acc_double))
    if rc != 0:
        raise RuntimeError("hrnet_oracle_forward failed (missing tensor)")
    return out


def decode_heatmaps(heatmaps: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    lib = _load()
    hm = np.ascontiguousarray(heatmaps, dtype=np.float32)
    n, j, h, w = hm.shape
    if boxes.dtype == np.int32:
        b, is_int = np.ascontiguousarray(boxes), 1
    else:
        b, is_int = np.ascontiguousarray(boxes, dtype=np.float32), 0
    pts = np.empty((n, j, 3), dtype=np.float32)
    lib.hrnet_oracle_decode(_fp(hm), n, j, h, w, b.ctypes.data_as(ctypes.c_void_p), is_int, _fp(pts))
    return pts
