"""simple-hrnet_amd -- MI355X-native (gfx950 / CDNA4) HRNet pose-inference hot path.

Drop-in for the model call + heat-map decode behind ``SimpleHRNet.predict()`` of
stefanopini/simple-HRNet (SimpleHRNet.py:281-308), plus its crop pre-path, flip-TTA evaluation decode, PoseResNet
and box NMS: hand-written HIP kernels behind the C ABI in
``include/hrnet_mi355.h``; this package is the thin ctypes / torch-tensor shim around it.

The directory name contains a hyphen (the name the project mandates), so import it with
``importlib.import_module("simple-hrnet_amd")`` or through the ``simple_hrnet_amd`` alias module
at the repository root.
"""
from . import _lib  # noqa: F401
from . import synth  # noqa: F401
from .native import NativeHRNet  # noqa: F401
from .nms import gpu_nms  # noqa: F401
from . import postproc  # noqa: F401
from .postproc import find_person_id_associations, oks_nms, soft_oks_nms  # noqa: F401
from .simple_hrnet import SimpleHRNet  # noqa: F401
from .synth import synth_boxes, synth_crops, synth_state_dict  # noqa: F401

__all__ = ["NativeHRNet", "SimpleHRNet", "gpu_nms", "oks_nms", "soft_oks_nms", "find_person_id_associations", "postproc", "synth", "synth_state_dict", "synth_crops", "synth_boxes"]
