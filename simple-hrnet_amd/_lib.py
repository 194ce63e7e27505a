"""ctypes binding of ``libhrnet_mi355.so`` (C ABI in ``include/hrnet_mi355.h``) and its build recipe.

The library is compiled in-tree with ``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU);
the resulting ``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
There is deliberately no fallback: if the library cannot be built or loaded, importing users fail.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libhrnet_mi355.so")
if os.environ.get("HRN_LIB_TAG"):   # A/B runs of compile-time variants (tools/mkvariant.sh builds libhrnet_mi355_<tag>.so beforehand)
    LIB_PATH = LIB_PATH.replace(".so", "_%s.so" % os.environ["HRN_LIB_TAG"])
SOURCES = ["kernels.hip", "conv3x3_lds.hip", "conv_s2.hip", "stem_fused.hip", "conv3x3_f32.hip", "bottleneck_chain.hip", "prepath.hip", "nms.hip", "postproc.cpp", "hrnet_mi355.cpp"]
HEADERS = [os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "conv3x3_n96.inc"), os.path.join(INCLUDE, "hrnet_mi355.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-Wno-inline-asm"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X HRNet library cannot be built")


def _obj_dir() -> str:
    """objects live beside the library they are linked into (a tagged variant -- other -D flags -- has its own)"""
    return LIB_PATH[:-3] + "_obj"


def _deps(src: str) -> List[str]:
    """files whose change invalidates the object of `src` (every source includes kernels.h; conv3x3_lds.hip the .inc files)"""
    d = [os.path.join(CSRC, src)] + HEADERS
    return d + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".inc") or f.endswith(".h")]


def _obj(src: str) -> str:
    return os.path.join(_obj_dir(), src.rsplit(".", 1)[0] + ".o")


def _stamp() -> str:
    """cached per flag set: `hipcc --version` is a subprocess, and needs_build() / _stale() ask once per source"""
    return _stamp_of(tuple(HIPCC_FLAGS))


import functools  # noqa: E402


@functools.lru_cache(maxsize=None)
def _stamp_of(flags) -> str:
    """what the objects of this library were compiled WITH: the flags (a tagged variant adds -D flags) and the compiler's version
    (objects built under other flags must not be reused because their sources are older)"""
    import hashlib
    try:
        ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:  # noqa: BLE001 -- no compiler: needs_build() decides whether that matters
        ver = "?"
    return hashlib.sha256((" ".join(flags) + "\n" + ver).encode()).hexdigest()


def _stamp_ok() -> bool:
    p = os.path.join(_obj_dir(), "build.stamp")
    return os.path.exists(p) and open(p).read().strip() == _stamp()


def _stale(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o) or not _stamp_ok():
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(src))


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if os.environ.get("HRN_LIB_TAG"):
        return False                  # a tagged variant is used as built
    t = os.path.getmtime(LIB_PATH)
    deps = set()
    for s in SOURCES:
        deps.update(_deps(s))
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # a library newer than its sources but built under other flags / another compiler, or whose objects are gone, is stale too --
    # unless there is no compiler to rebuild it with (the GPU box runs the prebuilt library that travelled with the snapshot)
    if os.path.isdir(_obj_dir()) and not _stamp_ok():
        return shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source of the package for gfx950 into ``libhrnet_mi355.so``: one object per source (only the
    stale ones, in parallel), then one link.

    Safe under concurrent callers (the N ranks of a torch.distributed launch all import the package): the build
    runs under an exclusive file lock, into per-process temporaries, and is skipped by whoever arrives second."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    from concurrent.futures import ThreadPoolExecutor

    # (the lock lives with the objects, not beside the library: nine stale `*.so.lock` files used to ship with every push)
    os.makedirs(_obj_dir(), exist_ok=True)
    with open(os.path.join(_obj_dir(), "build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                os.makedirs(_obj_dir(), exist_ok=True)
                cflags = [f for f in HIPCC_FLAGS if f != "-shared"]

                def compile_one(src):
                    tmp = "%s.tmp.%d" % (_obj(src), os.getpid())
                    cmd = [_hipcc()] + cflags + ["-c", "-o", tmp, os.path.join(CSRC, src)]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    subprocess.check_call(cmd)
                    os.replace(tmp, _obj(src))

                todo = [s for s in SOURCES if force or _stale(s)]
                stamp = _stamp()
                with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as ex:
                    list(ex.map(compile_one, todo))
                tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
                cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [_obj(s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                with open(os.path.join(_obj_dir(), "build.stamp"), "w") as f:
                    f.write(stamp + "\n")
                os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class TensorDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("dims", ctypes.c_int64 * 4), ("dtype", ctypes.c_int32)]


class ConvInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96),
                ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("ksize", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("relu", ctypes.c_int32), ("has_residual", ctypes.c_int32),
                ("in_h", ctypes.c_int32), ("in_w", ctypes.c_int32), ("out_h", ctypes.c_int32),
                ("out_w", ctypes.c_int32), ("kpad", ctypes.c_int32), ("nr", ctypes.c_int32),
                ("algo", ctypes.c_int32), ("ks", ctypes.c_int32),
                ("w_offset", ctypes.c_int64), ("w_bytes", ctypes.c_int64), ("b_offset", ctypes.c_int64),
                ("flops", ctypes.c_double)]


class TapInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("c", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
                ("conv_index", ctypes.c_int32)]


# every symbol include/hrnet_mi355.h declares: (restype, argtypes)
_P = ctypes.c_void_p
SYMBOLS = {
    "hrn_create": (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "hrn_create_model": (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "hrn_destroy": (None, [_P]),
    "hrn_last_error": (ctypes.c_char_p, [_P]),
    "hrn_load_weights": (ctypes.c_int, [_P, ctypes.POINTER(TensorDesc), ctypes.c_int]),
    "hrn_weight_blob_bytes": (ctypes.c_int64, [_P]),
    "hrn_weight_blob_ptr": (_P, [_P]),
    "hrn_adopt_weights": (ctypes.c_int, [_P]),
    "hrn_weight_blob_read": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int64]),
    "hrn_forward": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P]),
    "hrn_resize_frames": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "hrn_preprocess_frame": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]),
    "hrn_forward_flip_tta": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]),
    "hrn_nms": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]),
    "hrn_nms_release": (ctypes.c_int, [ctypes.c_int]),
    "hrn_nms_last_error": (ctypes.c_char_p, []),
    "hrn_oks_nms": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_double, _P, ctypes.c_double]),
    "hrn_soft_oks_nms": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_double, _P, ctypes.c_double]),
    "hrn_pose_similarity": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P]),
    "hrn_assignment": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P]),
    "hrn_tap_count": (ctypes.c_int, [_P]),
    "hrn_get_tap_info": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(TapInfo)]),
    "hrn_forward_tap": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "hrn_conv_count": (ctypes.c_int, [_P]),
    "hrn_get_conv_info": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ConvInfo)]),
    "hrn_flops_per_crop": (ctypes.c_double, [_P]),
    "hrn_workspace_bytes": (ctypes.c_int64, [_P]),
    "hrn_map_rebuilds": (ctypes.c_int64, [_P]),
    "hrn_launches_per_pass": (ctypes.c_int, [_P]),
    "hrn_switches": (ctypes.c_char_p, [_P]),
    "hrn_stem_fused": (ctypes.c_int, [_P]),
    "hrn_conv_compact": (ctypes.c_int, [_P, ctypes.c_int]),
    "hrn_debug_pad_violations": (ctypes.c_int64, [_P]),
    "hrn_plan_block_map": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int]),
    "hrn_plan_direct_map": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P]),
    "hrn_plan_s2_map": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P]),
    "hrn_profile_pass": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_float), _P]),
    "hrn_version": (ctypes.c_char_p, []),
}

_lib = None


def load(auto_build: bool = True) -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if auto_build and needs_build():
        build()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def header_symbols() -> List[str]:
    """names of the functions declared in include/hrnet_mi355.h (parsed, for the export test)"""
    import re

    text = open(os.path.join(INCLUDE, "hrnet_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hrn_[a-z_0-9]+)\s*\(", text)))
