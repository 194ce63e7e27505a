"""Host-side mirror of the reference's model seam, backed by the C ABI (include/hrnet_mi355.h).

``NativeHRNet`` is assignable to ``SimpleHRNet.model`` -- the reference invokes it as
``self.model(images)`` under ``torch.no_grad()`` and consumes ``.detach().cpu().numpy()``
(SimpleHRNet.py:284-296, 419-431), exactly as it does for its own TensorRT swap
(SimpleHRNet.py:143-147) -- and additionally offers the fused ``predict_crops`` (model call +
arg-max decode on the GPU, replacing SimpleHRNet.py:281-308 / 416-443).

PyTorch is used only as the container for device memory and for the current HIP stream.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib

DTYPES = {"fp32": 0, "f32": 0, "float32": 0, torch.float32: 0, "bf16": 1, "bfloat16": 1, torch.bfloat16: 1}


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError("Wrong device name.")  # same message as SimpleHRNet.py:139
    return d.index if d.index is not None else torch.cuda.current_device()


class NativeHRNet:
    """HRNet-W{c} pose network compiled to hand-written gfx950 kernels.

    Parameters mirror ``HRNet(c, nof_joints)`` (models_/hrnet.py:75) plus what a static engine
    needs up front: input ``resolution`` (h, w), arithmetic ``dtype`` ('bf16' MFMA with fp32
    accumulate, or 'fp32' exact-fp32 MFMA), and ``max_batch`` = crops per internal pass.
    ``device=-1`` builds a plan-only handle (graph + weight packing on the host, no GPU): it cannot
    run -- there is no CPU compute path.
    """

    def __init__(self, c: int = 48, nof_joints: int = 17, resolution: Tuple[int, int] = (384, 288),
                 dtype: Union[str, torch.dtype] = "bf16", max_batch: int = 32, device=0, model_name: str = "HRNet"):
        if dtype not in DTYPES:
            raise ValueError("dtype must be 'bf16' or 'fp32'")
        if model_name in ("HRNet", "hrnet"):            # SimpleHRNet.py:109-112
            model = 0
        elif model_name in ("PoseResNet", "poseresnet", "ResNet", "resnet"):
            model = 1                                  # c = ResNet size (50 / 101 / 152)
        else:
            raise ValueError("Wrong model name.")
        self.model_name = "HRNet" if model == 0 else "PoseResNet"
        self.c, self.nof_joints = int(c), int(nof_joints)
        self.resolution = (int(resolution[0]), int(resolution[1]))
        self.dtype = "bf16" if DTYPES[dtype] == 1 else "fp32"
        self.max_batch = int(max_batch)
        self.device_index = -1 if (isinstance(device, int) and device < 0) else _device_index(device)
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        rc = self._lib.hrn_create_model(ctypes.byref(self._h), model, self.c, self.nof_joints, self.resolution[0],
                                        self.resolution[1], DTYPES[dtype], self.max_batch, self.device_index)
        if rc != 0:
            msg = self._lib.hrn_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise (ValueError if rc == 2 else RuntimeError)("hrn_create failed: " + msg)
        self._keep = None

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hrn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self._lib.hrn_last_error(self._h).decode()))

    # -- nn.Module look-alikes the reference calls on self.model (SimpleHRNet.py:141-142) --------
    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    @property
    def torch_device(self) -> torch.device:
        return torch.device("cuda", self.device_index)

    # -- weights --------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict) -> "NativeHRNet":
        """Accepts what ``torch.load(checkpoint)`` yields: a raw ``state_dict`` or ``{'model': ...}``
        (SimpleHRNet.py:117-121); values may be torch tensors or numpy arrays."""
        if "model" in state_dict and not hasattr(state_dict["model"], "shape"):
            state_dict = state_dict["model"]
        keep, descs = [], []
        for k, v in state_dict.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            a = np.asarray(v)
            if a.dtype == np.int64:
                code = 1
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                code = 0
            if a.ndim > 4:
                raise ValueError("state_dict entry %s has %d dims" % (k, a.ndim))
            keep.append(a)
            dims = (ctypes.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim)))
            descs.append(_lib.TensorDesc(k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.ndim, dims, code))
        arr = (_lib.TensorDesc * len(descs))(*descs)
        rc = self._lib.hrn_load_weights(self._h, arr, len(descs))
        if rc != 0:
            raise KeyError("hrn_load_weights: " + self._lib.hrn_last_error(self._h).decode())
        return self

    def load_checkpoint(self, path: str) -> "NativeHRNet":
        return self.load_state_dict(torch.load(path, map_location="cpu"))

    # -- multi-GPU weight distribution (RCCL broadcast of the packed blob) -------------------------
    def weight_blob_bytes(self) -> int:
        return int(self._lib.hrn_weight_blob_bytes(self._h))

    def weight_blob_tensor(self) -> torch.Tensor:
        """uint8 CUDA tensor aliasing the packed-weight blob (zero-copy, for torch.distributed)."""

        class _Raw:
            pass

        raw = _Raw()
        raw.__cuda_array_interface__ = {"shape": (self.weight_blob_bytes(),), "typestr": "|u1",
                                        "data": (int(self._lib.hrn_weight_blob_ptr(self._h)), False), "version": 2}
        raw._owner = self
        with torch.cuda.device(self.device_index):
            return torch.as_tensor(raw, device=self.torch_device)

    def adopt_weights(self):
        self._check(self._lib.hrn_adopt_weights(self._h), "hrn_adopt_weights")

    def read_blob(self, offset: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, dtype=np.uint8)
        self._check(self._lib.hrn_weight_blob_read(self._h, offset, out.ctypes.data_as(ctypes.c_void_p), nbytes),
                    "hrn_weight_blob_read")
        return out

    # -- the hot path ---------------------------------------------------------------------------
    def _images_ptr(self, images: torch.Tensor) -> torch.Tensor:
        if not isinstance(images, torch.Tensor):
            raise TypeError("images must be a torch.Tensor (n,3,H,W)")
        h, w = self.resolution
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, h, w):
            raise ValueError("images must have shape (n,3,%d,%d), got %s" % (h, w, tuple(images.shape)))
        if images.device.type != "cuda":
            images = images.to(self.torch_device, non_blocking=True)
        elif images.device.index != self.device_index:
            raise ValueError("images live on %s, engine on cuda:%d" % (images.device, self.device_index))
        return images.to(torch.float32).contiguous()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device_index).cuda_stream

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        """``self.model(images)``: (n,3,H,W) fp32 -> heat-maps (n,J,H/4,W/4) fp32 on the same GPU."""
        x = self._images_ptr(images)
        n = x.shape[0]
        h, w = self.resolution
        out = torch.empty((n, self.nof_joints, h // 4, w // 4), dtype=torch.float32, device=x.device)
        if n:
            with torch.cuda.device(self.device_index):
                self._check(self._lib.hrn_forward(self._h, x.data_ptr(), n, None, 0, None, out.data_ptr(),
                                                  self._stream()), "hrn_forward")
        return out

    forward = __call__

    def predict_crops(self, images: torch.Tensor, boxes, return_heatmaps: bool = False):
        """Model call + decode (SimpleHRNet.py:281-308): returns ``pts`` (n,J,3) fp32 CUDA tensor of
        ``(y, x, confidence)``; with ``return_heatmaps`` also the (n,J,H/4,W/4) heat-maps.
        ``boxes``: (n,4) ``[x1,y1,x2,y2]`` int32 (multi-person path) or float32 (single-person)."""
        x = self._images_ptr(images)
        n = x.shape[0]
        b = boxes if isinstance(boxes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(boxes))
        if tuple(b.shape) != (n, 4):
            raise ValueError("boxes must have shape (%d,4), got %s" % (n, tuple(b.shape)))
        if b.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8):
            b, box_dtype = b.to(torch.int32), 0
        else:
            b, box_dtype = b.to(torch.float32), 1
        b = b.to(x.device).contiguous()
        h, w = self.resolution
        pts = torch.empty((n, self.nof_joints, 3), dtype=torch.float32, device=x.device)
        hm = torch.empty((n, self.nof_joints, h // 4, w // 4), dtype=torch.float32,
                         device=x.device) if return_heatmaps else None
        if n:
            with torch.cuda.device(self.device_index):
                self._check(self._lib.hrn_forward(self._h, x.data_ptr(), n, b.data_ptr(), box_dtype, pts.data_ptr(),
                                                  hm.data_ptr() if hm is not None else None, self._stream()),
                            "hrn_forward")
        return (hm, pts) if return_heatmaps else pts

    def predict_stream(self, batches, return_heatmaps: bool = False):
        """Model call + decode for a sequence of HOST-resident batches with the uploads hidden behind the compute:
        batch k+1 crosses PCIe on a copy stream (two device staging buffers) while batch k runs on the current stream.
        ``batches``: iterable of ``(images, boxes (n,4))`` with ``n <= max_batch``; ``images`` is a HOST tensor (pinned memory for
        a truly asynchronous copy), either ``(n,3,H,W) float32`` -- the normalised crops the reference builds on the CPU
        (``SimpleHRNet.py:213-232``) -- or ``(n,H,W,3) uint8`` BGR crops already at the network's resolution (what ``cv2.resize``
        leaves before ``cvtColor`` / ``ToTensor`` / ``Normalize``): a quarter of the bytes over PCIe (85 MB instead of 340 MB per
        256 crops of 384x288), the colour flip and the normalisation then run on the GPU (``hrn_resize_frames`` at identity size:
        the reference transform's float32 arithmetic).  Yields, per batch, what ``predict_crops`` returns (results of batch k
        are ready on the current stream; read them after a synchronize or through ``.cpu()``)."""
        dev = self.torch_device
        compute = torch.cuda.current_stream(dev)
        copy = torch.cuda.Stream(dev)
        h, w = self.resolution
        stage = [None, None]   # allocated for the first batch: fp32 NCHW or uint8 NHWC staging, by what the caller sends
        # the staging blocks come from the caching allocator on the COMPUTE stream and may be recycled memory that
        # kernels already queued there still read: the first upload must not overtake them
        copy.wait_stream(compute)
        landed = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [None, None]
        norm = [None, None]   # uint8 form: the normalised fp32 crops of each slot

        def upload(slot, item):
            images, boxes = item
            if not isinstance(images, torch.Tensor) or images.device.type != "cpu":
                raise TypeError("predict_stream takes host tensors; device-resident crops go to predict_crops")
            n = images.shape[0]
            if n > self.max_batch:
                raise ValueError("a batch of %d crops exceeds max_batch=%d" % (n, self.max_batch))
            u8 = images.dtype == torch.uint8
            if u8 and tuple(images.shape[1:]) != (h, w, 3):
                raise ValueError("uint8 crops must be (n, %d, %d, 3) BGR at the network's resolution" % (h, w))
            if not u8 and tuple(images.shape[1:]) != (3, h, w):   # (copy_ would broadcast or fail obscurely)
                raise ValueError("float crops must be (n, 3, %d, %d) at the network's resolution" % (h, w))
            want = (torch.uint8, (self.max_batch, h, w, 3)) if u8 else (torch.float32, (self.max_batch, 3, h, w))
            if stage[slot] is None or stage[slot].dtype != want[0]:
                # a staging block comes from the caching allocator on the COMPUTE stream and may be recycled memory that kernels already
                # queued there still read: the upload into it must not overtake them
                stage[slot] = torch.empty(want[1], dtype=want[0], device=dev)
                stage[slot].record_stream(copy)
                copy.wait_stream(compute)
            with torch.cuda.stream(copy):
                if consumed[slot] is not None:
                    copy.wait_event(consumed[slot])              # the pass that read this buffer has finished
                stage[slot][:n].copy_(images if u8 else images.to(torch.float32), non_blocking=True)
                landed[slot].record(copy)
            return n, boxes, u8

        it = iter(batches)
        nxt = next(it, None)
        pending = upload(0, nxt) if nxt is not None else None
        k = 0
        while pending is not None:
            slot = k & 1
            n, boxes, u8 = pending
            nxt = next(it, None)
            pending = upload(slot ^ 1, nxt) if nxt is not None else None   # goes out while this batch computes
            compute.wait_event(landed[slot])
            if u8:   # identity size: colour flip + ToTensor + Normalize on the GPU, into ONE fp32 buffer per slot (340 MB at 256 crops)
                if norm[slot] is None:
                    norm[slot] = torch.empty((self.max_batch, 3, h, w), dtype=torch.float32, device=dev)
                x = self.resize_frames(stage[slot][:n], 0, out=norm[slot])
            else:
                x = stage[slot][:n]
            out = self.predict_crops(x, boxes, return_heatmaps=return_heatmaps)
            consumed[slot] = torch.cuda.Event()
            consumed[slot].record(compute)
            yield out
            k += 1

    # -- introspection --------------------------------------------------------------------------
    def predict_flip_tta(self, images: torch.Tensor, flip_pairs, post_processing: bool = True):
        """Flip test-time augmentation + evaluation decode (``testing/Test.py:132-140``, ``misc/utils.py:9-29, 125-175``):
        returns ``(heatmaps (n,J,h,w) averaged over the crop and its mirror image, preds (n,J,2) = (x, y) in heat-map
        pixels with the quarter-pixel refinement, maxvals (n,J,1))`` -- what ``get_final_preds`` works on before its
        inverse affine."""
        x = self._images_ptr(images)
        n = x.shape[0]
        fp = np.ascontiguousarray(np.asarray(flip_pairs, dtype=np.int32).reshape(-1, 2))
        h, w = self.resolution
        hm = torch.empty((n, self.nof_joints, h // 4, w // 4), dtype=torch.float32, device=x.device)
        preds = torch.empty((n, self.nof_joints, 2), dtype=torch.float32, device=x.device)
        maxvals = torch.empty((n, self.nof_joints, 1), dtype=torch.float32, device=x.device)
        if n:
            with torch.cuda.device(self.device_index):
                self._check(self._lib.hrn_forward_flip_tta(self._h, x.data_ptr(), n, fp.ctypes.data, len(fp),
                                                           1 if post_processing else 0, hm.data_ptr(), preds.data_ptr(),
                                                           maxvals.data_ptr(), self._stream()), "hrn_forward_flip_tta")
        return hm, preds, maxvals

    def preprocess_frame(self, frame: torch.Tensor, detections, variant: str = "pad") -> Tuple[torch.Tensor, np.ndarray, torch.Tensor]:
        """The crop pre-path of ``SimpleHRNet.predict`` for one frame (``SimpleHRNet.py:236-278``) on the GPU.

        ``frame``: (Hf, Wf, 3) uint8 BGR (the cv2 frame; host tensors / arrays are uploaded once);
        ``detections``: (P, >=4) float array-like, columns 0..3 = x1, y1, x2, y2 as the detector returns them.
        ``variant``: ``"pad"`` = the single-image path (aspect ratio corrected by zero padding), ``"clamp"`` = the batch
        path's enlarge-and-clamp (``SimpleHRNet.py:383-412``; call once per image of the stack).
        Returns ``(images (P,3,H,W) float32 on the GPU, boxes (P,4) int32 numpy, boxes on the GPU)`` -- bit-identical
        to the reference's ``ToPILImage -> Resize -> ToTensor -> Normalize`` of the RGB crops."""
        if variant not in ("pad", "clamp"):
            raise ValueError("variant must be 'pad' or 'clamp'")
        if not isinstance(frame, torch.Tensor):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise ValueError("frame must be (H, W, 3) uint8 BGR")
        frame = frame.to(self.torch_device, non_blocking=True).contiguous()
        dets = np.ascontiguousarray(np.asarray(detections.cpu() if isinstance(detections, torch.Tensor) else detections,
                                               dtype=np.float32))
        if dets.ndim != 2 or (len(dets) and dets.shape[1] < 4):
            raise ValueError("detections must be (P, >=4)")
        p = len(dets)
        h, w = self.resolution
        images = torch.empty((p, 3, h, w), dtype=torch.float32, device=self.torch_device)
        boxes = np.empty((p, 4), dtype=np.int32)
        boxes_dev = torch.empty((p, 4), dtype=torch.int32, device=self.torch_device)
        if p:
            with torch.cuda.device(self.device_index):
                rc = self._lib.hrn_preprocess_frame(self._h, frame.data_ptr(), int(frame.shape[0]), int(frame.shape[1]),
                                                    dets.ctypes.data, int(dets.shape[1]), p, 0 if variant == "pad" else 1,
                                                    images.data_ptr(),
                                                    boxes.ctypes.data, boxes_dev.data_ptr(), self._stream())
            self._check(rc, "hrn_preprocess_frame")
        return images, boxes, boxes_dev

    def resize_frames(self, frames, interpolation: int = 2, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The single-person pre-path (``multiperson=False``, ``SimpleHRNet.py:213-222`` / ``:355-366``) on the GPU:
        ``cv2.resize(frame, (W, H), interpolation)`` + BGR -> RGB + ToTensor + Normalize for every frame.

        ``frames``: (Hf, Wf, 3) or (n, Hf, Wf, 3) uint8 BGR (host arrays are uploaded once); ``interpolation``: the
        ``cv2.INTER_*`` value -- 0 nearest, 1 linear, 2 cubic (the reference's default).  Returns (n, 3, H, W) float32 on the GPU
        (written into the first n entries of ``out`` when given: ``predict_stream`` keeps one such buffer per staging slot).
        Follows OpenCV's published generic 8-bit path; equality with a particular cv2 build is not pinned (include/hrnet_mi355.h)."""
        if interpolation not in (0, 1, 2):
            raise ValueError("interpolation must be cv2.INTER_NEAREST (0), cv2.INTER_LINEAR (1) or cv2.INTER_CUBIC (2)")
        if not isinstance(frames, torch.Tensor):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dim() == 3:
            frames = frames.unsqueeze(0)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be (n, H, W, 3) uint8 BGR")
        frames = frames.to(self.torch_device, non_blocking=True).contiguous()
        n, h, w = int(frames.shape[0]), *self.resolution
        if out is not None:
            if out.dtype != torch.float32 or out.device != self.torch_device or not out.is_contiguous() or \
                    out.dim() != 4 or out.shape[0] < n or tuple(out.shape[1:]) != (3, h, w):
                raise ValueError("out must be a contiguous (>= %d, 3, %d, %d) float32 tensor on %s" % (n, h, w, self.torch_device))
            images = out[:n]
        else:
            images = torch.empty((n, 3, h, w), dtype=torch.float32, device=self.torch_device)
        if n:
            with torch.cuda.device(self.device_index):
                rc = self._lib.hrn_resize_frames(self._h, frames.data_ptr(), n, int(frames.shape[1]), int(frames.shape[2]),
                                                 int(interpolation), images.data_ptr(), self._stream())
            self._check(rc, "hrn_resize_frames")
        return images

    def predict_frame(self, frame, detections, return_heatmaps: bool = False, variant: str = "pad"):
        """pre-path + model + decode for one frame: what ``SimpleHRNet._predict_single`` does after the detector.
        Returns ``(boxes (P,4) int32 numpy, pts (P,J,3) on the GPU[, heatmaps])``."""
        images, boxes, boxes_dev = self.preprocess_frame(frame, detections, variant)
        out = self.predict_crops(images, boxes_dev, return_heatmaps=return_heatmaps)
        if return_heatmaps:
            return boxes, out[1], out[0]
        return boxes, out

    def tap_infos(self) -> List[_lib.TapInfo]:
        """the tensors ``forward_tap`` can read: name, (c, h, w), index of the convolution that writes it (or -1)"""
        out = []
        for i in range(self._lib.hrn_tap_count(self._h)):
            ti = _lib.TapInfo()
            self._check(self._lib.hrn_get_tap_info(self._h, i, ctypes.byref(ti)), "hrn_get_tap_info")
            out.append(ti)
        return out

    def forward_tap(self, images: torch.Tensor, name: str, crop0: int = 0, ncrops: Optional[int] = None,
                    return_heatmaps: bool = False, crop_step: int = 1):
        """Debug tap (the engine's forward hook): ONE micro-batch over ``images`` (n <= max_batch) with the named
        intermediate tensor of crops ``crop0, crop0 + crop_step, ...`` (``ncrops`` of them) copied out as (ncrops, C, H, W) fp32 -- for the bf16
        engine the stored bf16 values, widened exactly.  Names: ``tap_infos()`` / include/hrnet_mi355.h."""
        x = self._images_ptr(images)
        n = x.shape[0]
        if ncrops is None:
            ncrops = (n - crop0 + crop_step - 1) // crop_step
        info = {t.name.decode(): t for t in self.tap_infos()}.get(name)
        if info is None:
            raise KeyError("no tensor named %r is written by this plan" % name)
        dst = torch.empty((ncrops, info.c, info.h, info.w), dtype=torch.float32, device=x.device)
        h, w = self.resolution
        hm = torch.empty((n, self.nof_joints, h // 4, w // 4), dtype=torch.float32, device=x.device) if return_heatmaps else None
        with torch.cuda.device(self.device_index):
            self._check(self._lib.hrn_forward_tap(self._h, x.data_ptr(), n, name.encode(), int(crop0), int(ncrops), int(crop_step),
                                                  dst.data_ptr(), hm.data_ptr() if hm is not None else None, self._stream()),
                        "hrn_forward_tap")
        return (dst, hm) if return_heatmaps else dst

    def conv_infos(self) -> List[_lib.ConvInfo]:
        out = []
        for i in range(self._lib.hrn_conv_count(self._h)):
            ci = _lib.ConvInfo()
            self._check(self._lib.hrn_get_conv_info(self._h, i, ctypes.byref(ci)), "hrn_get_conv_info")
            out.append(ci)
        return out

    def flops_per_crop(self) -> float:
        return float(self._lib.hrn_flops_per_crop(self._h))

    def workspace_bytes(self) -> int:
        return int(self._lib.hrn_workspace_bytes(self._h))

    def map_rebuilds(self) -> int:
        """block maps / descriptor arrays built and uploaded so far (they depend on the micro-batch size only)"""
        return int(self._lib.hrn_map_rebuilds(self._h))

    def launches_per_pass(self) -> int:
        return int(self._lib.hrn_launches_per_pass(self._h))

    def conv_compact(self, index: int) -> bool:
        """convolution `index` (conv_infos numbering) runs the 96-cout form over real pixels only (csrc/conv3x3_n96.inc, CP)"""
        return bool(self._lib.hrn_conv_compact(self._h, index))

    def stem_fused(self) -> bool:
        """conv1 + conv2 of the stem run as one kernel (csrc/stem_fused.hip)"""
        return bool(self._lib.hrn_stem_fused(self._h))

    def pad_violations(self) -> int:
        """debug: non-zero elements at pad / guard positions of the activation workspace (0 = the layout's invariant holds)"""
        return int(self._lib.hrn_debug_pad_violations(self._h))

    def switches(self) -> str:
        """the HRN_* environment switches this engine saw when it was created (always "" unless the process opted in with HRN_DEBUG_ENV=1)"""
        return self._lib.hrn_switches(self._h).decode()

    def profile_pass(self, images: torch.Tensor):
        """HIP-event time of every kernel of ONE internal pass over ``images[:max_batch]``.
        Returns (conv_ms[list], other_ms{stem,fuse,head,decode})."""
        x = self._images_ptr(images)
        n = min(x.shape[0], self.max_batch)
        nconv = self._lib.hrn_conv_count(self._h)
        conv_ms = (ctypes.c_float * nconv)()
        other = (ctypes.c_float * 4)()
        with torch.cuda.device(self.device_index):
            self._check(self._lib.hrn_profile_pass(self._h, x.data_ptr(), n, conv_ms, nconv, other, self._stream()),
                        "hrn_profile_pass")
        return list(conv_ms), dict(zip(("stem", "fuse", "head", "decode"), list(other)))


class MultiDeviceHRNet:
    """One process, several GPUs -- what ``torch.nn.DataParallel(model, device_ids)`` gives a user of the reference
    with ONE call (``SimpleHRNet.py:123-135``), without its per-forward parameter broadcast and heat-map gather:

    * one ``NativeHRNet`` handle per listed device (a device may be listed twice: two handles, two streams -- "lanes" of one
      GPU: each lane runs half the batch, and the launches of one fill the drain / tail of the other's, +4 % on a
      256-crop W48 batch, same joints);
    * weights are folded + packed once (first handle) and copied blob-to-blob to the others;
    * ``predict_crops`` / ``__call__`` / ``predict_frame`` split the packed crop batch (or the frame's detections) into
      contiguous index ranges (``dist.shard_range``), run every range on its device from its own host thread (the C ABI
      releases the GIL) on a per-handle side stream, and gather the small results on the first device.

    Same method surface as ``NativeHRNet`` for what ``SimpleHRNet`` uses.  Multi-process jobs (one rank per GPU,
    ``dist.ShardedHRNet``) remain the way to scale a serving loop; this class is the drop-in for the reference's
    single-process ``device='cuda'`` / ``'cuda:1,2'``."""

    def __init__(self, devices: Sequence[int], c: int = 48, nof_joints: int = 17, resolution: Tuple[int, int] = (384, 288),
                 dtype="bf16", max_batch: int = 32, model_name: str = "HRNet"):
        from concurrent.futures import ThreadPoolExecutor

        if len(devices) < 1:
            raise ValueError("at least one device")
        self.devices = [int(d) for d in devices]
        self.nets = [NativeHRNet(c, nof_joints, resolution, dtype, max_batch=max_batch, device=d, model_name=model_name)
                     for d in self.devices]
        self.streams = [torch.cuda.Stream(torch.device("cuda", d)) for d in self.devices]
        self.pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="hrn-dev")
        first = self.nets[0]
        self.c, self.nof_joints, self.resolution = first.c, first.nof_joints, first.resolution
        self.dtype, self.max_batch, self.model_name = first.dtype, first.max_batch, first.model_name
        self.device_index = first.device_index

    @property
    def torch_device(self) -> torch.device:
        return self.nets[0].torch_device

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def close(self):
        for n in getattr(self, "nets", []):
            n.close()
        pool = getattr(self, "pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights: fold + pack once, then blob -> blob (device to device; peer copy across GPUs) -----------------------
    def load_state_dict(self, state_dict: Dict) -> "MultiDeviceHRNet":
        self.nets[0].load_state_dict(state_dict)
        src = self.nets[0].weight_blob_tensor()
        for net in self.nets[1:]:
            net.weight_blob_tensor().copy_(src)
            torch.cuda.synchronize(net.torch_device)
            net.adopt_weights()
        torch.cuda.synchronize(self.torch_device)
        return self

    def load_checkpoint(self, path: str) -> "MultiDeviceHRNet":
        return self.load_state_dict(torch.load(path, map_location="cpu"))

    def adopt_from(self, src: "NativeHRNet") -> "MultiDeviceHRNet":
        """weights from an engine that already holds them (loaded, or received over RCCL): blob -> blob, no re-packing"""
        blob = src.weight_blob_tensor()
        for net in self.nets:
            net.weight_blob_tensor().copy_(blob)
            torch.cuda.synchronize(net.torch_device)
            net.adopt_weights()
        return self

    # -- sharded execution ---------------------------------------------------------------------------------------------
    def _ranges(self, n: int) -> List[Tuple[int, int]]:
        from .dist import shard_range

        return [shard_range(n, len(self.nets), k) for k in range(len(self.nets))]

    def _run(self, n: int, work):
        """work(k, net, lo, hi) -> tuple of tensors on net's device (or None); runs every non-empty range on its own
        thread and stream, orders it after the caller's stream on every device involved, returns the per-range results
        moved to the first device (in range order)."""
        first = self.torch_device
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(first))   # inputs produced on the caller's stream of the first device

        def job(k, lo, hi):
            net, stream = self.nets[k], self.streams[k]
            with torch.cuda.device(net.device_index), torch.cuda.stream(stream):
                stream.wait_event(ready)
                out = work(k, net, lo, hi)
                out = tuple(t if t is None or not isinstance(t, torch.Tensor) else t.to(first, non_blocking=True) for t in out)
                done = torch.cuda.Event()
                done.record(stream)
                return out, done

        futs = [(self.pool.submit(job, k, lo, hi)) for k, (lo, hi) in enumerate(self._ranges(n)) if hi > lo]
        outs = []
        mine = torch.cuda.current_stream(first)
        for f in futs:
            out, done = f.result()
            mine.wait_event(done)
            for t in out:   # produced under a side stream's allocator pool, consumed on the caller's stream
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(mine)
            outs.append(out)
        return outs

    def resize_frames(self, frames, interpolation: int = 2) -> torch.Tensor:
        """single-person pre-path (``NativeHRNet.resize_frames``) on the first device; ``predict_crops`` shards the result"""
        return self.nets[0].resize_frames(frames, interpolation)

    def predict_crops(self, images: torch.Tensor, boxes, return_heatmaps: bool = False):
        n = int(images.shape[0])
        b = boxes if isinstance(boxes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(boxes))
        if n == 0:
            return self.nets[0].predict_crops(images, b, return_heatmaps=return_heatmaps)

        def work(k, net, lo, hi):
            x = images[lo:hi].to(net.torch_device, non_blocking=True)
            out = net.predict_crops(x, b[lo:hi], return_heatmaps=return_heatmaps)
            return out if return_heatmaps else (out,)

        outs = self._run(n, work)
        if return_heatmaps:
            return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)
        return torch.cat([o[0] for o in outs], 0)

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        n = int(images.shape[0])
        if n == 0:
            return self.nets[0](images)
        outs = self._run(n, lambda k, net, lo, hi: (net(images[lo:hi].to(net.torch_device, non_blocking=True)),))
        return torch.cat([o[0] for o in outs], 0)

    forward = __call__

    def preprocess_frame(self, frame, detections, variant: str = "pad"):
        """crops of one frame, produced on the devices that will run them and gathered on the first one"""
        dets = np.ascontiguousarray(np.asarray(detections.cpu() if isinstance(detections, torch.Tensor) else detections,
                                               dtype=np.float32))
        if len(dets) == 0:
            return self.nets[0].preprocess_frame(frame, dets, variant)
        if not isinstance(frame, torch.Tensor):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        boxes_np = [None] * len(self.nets)

        def work(k, net, lo, hi):
            images, bx, bx_dev = net.preprocess_frame(frame, dets[lo:hi], variant)
            boxes_np[k] = bx
            return images, bx_dev

        outs = self._run(len(dets), work)
        return (torch.cat([o[0] for o in outs], 0), np.concatenate([b for b in boxes_np if b is not None], 0),
                torch.cat([o[1] for o in outs], 0))

    def predict_frame(self, frame, detections, return_heatmaps: bool = False, variant: str = "pad"):
        dets = np.ascontiguousarray(np.asarray(detections.cpu() if isinstance(detections, torch.Tensor) else detections,
                                               dtype=np.float32))
        if len(dets) == 0:
            return self.nets[0].predict_frame(frame, dets, return_heatmaps=return_heatmaps, variant=variant)
        if not isinstance(frame, torch.Tensor):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        boxes_np = [None] * len(self.nets)

        def work(k, net, lo, hi):
            out = net.predict_frame(frame, dets[lo:hi], return_heatmaps=return_heatmaps, variant=variant)
            boxes_np[k] = out[0]
            return tuple(out[1:])

        outs = self._run(len(dets), work)
        boxes = np.concatenate([b for b in boxes_np if b is not None], 0)
        pts = torch.cat([o[0] for o in outs], 0)
        if return_heatmaps:
            return boxes, pts, torch.cat([o[1] for o in outs], 0)
        return boxes, pts
