"""``SimpleHRNet`` with the reference's constructor and ``predict()`` contract (``SimpleHRNet.py:21-172, 174-496``) on
top of the MI355X library -- what a user of the reference instantiates after switching.

Same argument names and meaning, same return structure (numpy arrays: heat-maps float32, boxes int32 / float32, joints
``(y, x, confidence)`` float32; lists per image for stacks), same error for a wrong model name or image rank.  What is
different, and why:

* the person detector is injected (``detector=``: an object with ``predict_single(image)`` / ``predict(images)``
  returning ``(P, >=4)`` rows ``x1, y1, x2, y2, ...`` or ``None``) -- the reference's YOLOv3 wrapper depends on an
  un-vendored third-party submodule (``models_/detectors/yolo``) and YOLOv5 on ``torch.hub`` (network);
* ``multiperson=False``: frames of another size than the model resolution are resized on the GPU the way
  ``cv2.resize(frame, (W, H), interpolation)`` does it (``:213-218``; ``interpolation`` = ``cv2.INTER_NEAREST`` 0 /
  ``INTER_LINEAR`` 1 / ``INTER_CUBIC`` 2, default cubic as in the reference, anything else raises) -- OpenCV's published generic
  8-bit arithmetic, NOT pinned against a cv2 build (cv2 is absent here; ``include/hrnet_mi355.h``: ``hrn_resize_frames``);
* ``dtype`` picks the arithmetic mode of the engine (``"fp32"`` = parity mode, ``"bf16"`` = MFMA bf16);
* devices: ``'cuda:N'`` is that GPU.  ``'cuda'`` (all GPUs) and ``'cuda:1,2'`` (the listed ones) are, in a plain Python
  process, ONE engine per listed GPU driven from this process (``native.MultiDeviceHRNet``: the crop batch of a
  ``predict()`` call is split by index range, one host thread per GPU) -- what ``DataParallel`` gives the reference with
  one call (``SimpleHRNet.py:123-135``); under ``torch.distributed.run`` (``LOCAL_RANK`` set) the same strings name the
  job's GPUs and each process takes the one of its rank (``resolve_device``), to be combined with ``dist.ShardedHRNet``.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .native import MultiDeviceHRNet, NativeHRNet

_MEAN = (0.485, 0.456, 0.406)   # SimpleHRNet.py:171
_STD = (0.229, 0.224, 0.225)


# (local-rank variable, task-count variables of the same launcher): srun / mpirun export their local rank even for a one-task
# job, so the variable only means "one GPU per process" when the job has more than one task
_LAUNCHERS = (("SLURM_LOCALID", ("SLURM_NTASKS", "SLURM_NPROCS")),
              ("OMPI_COMM_WORLD_LOCAL_RANK", ("OMPI_COMM_WORLD_SIZE",)),
              ("MV2_COMM_WORLD_LOCAL_RANK", ("MV2_COMM_WORLD_SIZE",)),
              ("MPI_LOCALRANKID", ("PMI_SIZE",)))


def _launcher_local_rank() -> Optional[int]:
    """the GPU slot a multi-process launcher gave this process, or None for a plain process.  ``torch.distributed.run`` sets
    LOCAL_RANK (taken as it is: torchrun does not restrict the visible devices).  srun / mpirun set their own variables and
    no LOCAL_RANK -- under any of them a process drives ONE GPU (a rank that built an engine on every visible GPU would
    oversubscribe the node) -- but only when the job really has several tasks, and the rank is folded onto the
    devices this task can SEE: under ``srun --gpus-per-task=1`` / gpu-bind every task sees a single GPU at index 0 while
    SLURM_LOCALID runs 0..7."""
    ndev = max(1, torch.cuda.device_count())
    if "LOCAL_RANK" in os.environ:
        # (taken as it is: whether it names a visible device is checked where it is USED as the device index -- _check_local_rank;
        # an explicit 'cuda:i[,j]' list maps it with a modulo instead)
        return int(os.environ["LOCAL_RANK"])
    for key, sizes in _LAUNCHERS:
        if key in os.environ:
            counts = [int(os.environ[k]) for k in sizes if os.environ.get(k, "").isdigit()]
            if not counts and os.environ.get("WORLD_SIZE", "").isdigit():
                counts = [int(os.environ["WORLD_SIZE"])]
            # one GPU per process unless the job is EXPLICITLY a single task (a launcher that exports a local rank
            # but no task count used to fall through and let every task build engines on all visible GPUs)
            if not counts or max(counts) > 1:
                return int(os.environ[key]) % ndev
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:   # a launcher without a local-rank variable
        return int(os.environ["RANK"]) % ndev
    return None


def _check_local_rank(lr: int) -> int:
    """ torchrun does not restrict the visible devices: a local rank used DIRECTLY as the device index (device
    None / 'cuda' / torch.device('cuda')) beyond them is a launch error, said here rather than as an invalid-device failure deep
    inside the engine.  Explicit lists ('cuda:0', 'cuda:0,0', one visible GPU per rank) never come through here."""
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if lr < 0 or (ndev and lr >= ndev):
        raise ValueError("LOCAL_RANK=%d but this process sees %d GPU(s): launch at most one rank per visible GPU" % (lr, ndev))
    return lr


def resolve_device(device, local_rank: Optional[int] = None) -> torch.device:
    """The reference's device argument (``SimpleHRNet.py:123-139``) mapped onto one-process-per-GPU: ``'cuda:3'`` is that
    GPU; ``'cuda'`` (all GPUs through DataParallel there) and ``'cuda:1,2'`` (the listed ones) name the set this job runs
    on, of which THIS process takes the entry of its ``LOCAL_RANK`` (0 when not launched by ``torch.distributed.run``).
    Anything that is not CUDA raises, as the reference does for a wrong name -- there is no CPU path here."""
    if local_rank is None:
        local_rank = _launcher_local_rank() or 0
    if device is None:
        return torch.device("cuda", _check_local_rank(local_rank))
    if isinstance(device, torch.device):
        if device.type != "cuda":
            raise ValueError("the MI355X engine has no CPU path (device=%s)" % device)
        return torch.device("cuda", _check_local_rank(local_rank) if device.index is None else device.index)
    name = str(device)
    if name == "cuda":
        return torch.device("cuda", _check_local_rank(local_rank))
    if name.startswith("cuda:"):
        try:
            ids = [int(x) for x in name[5:].split(",")]
        except ValueError:
            raise ValueError("Wrong device name.") from None           # SimpleHRNet.py:139
        if ids and all(i >= 0 for i in ids):
            return torch.device("cuda", ids[local_rank % len(ids)])
    raise ValueError("Wrong device name." if name.startswith("cuda") else "the MI355X engine has no CPU path (device=%s)" % name)


def resolve_devices(device) -> list:
    """GPU indices THIS process drives for the reference's ``device`` argument.  Launched by ``torch.distributed.run``
    or another multi-process launcher (``LOCAL_RANK`` / ``SLURM_LOCALID`` / ``OMPI_COMM_WORLD_LOCAL_RANK`` / ``RANK`` +
    ``WORLD_SIZE``): exactly one, ``resolve_device``'s.  A plain process: ``'cuda'`` = every visible GPU,
    ``'cuda:1,2'`` = those (an index may repeat), ``'cuda:3'`` / ``torch.device('cuda', 3)`` = that one, ``None`` = GPU 0."""
    if _launcher_local_rank() is not None:
        return [resolve_device(device).index]
    if device is None:
        return [0]
    if isinstance(device, torch.device):
        if device.type == "cuda" and device.index is None:
            return list(range(max(1, torch.cuda.device_count())))   # (no visible GPU: index 0, and the engine says so when created)
        return [resolve_device(device).index]
    name = str(device)
    if name == "cuda":
        return list(range(max(1, torch.cuda.device_count())))
    if name.startswith("cuda:") and "," in name:
        resolve_device(name)   # validates (raises the reference's error for a malformed list)
        return [int(x) for x in name[5:].split(",")]
    return [resolve_device(device, 0).index]


class SimpleHRNet:
    def __init__(self, c, nof_joints, checkpoint_path, model_name="HRNet", resolution=(384, 288), interpolation=None,
                 multiperson=True, return_heatmaps=False, return_bounding_boxes=False, max_batch_size=32,
                 yolo_version="v3", yolo_model_def=None, yolo_class_path=None, yolo_weights_path=None, device=None,
                 enable_tensorrt=False, *, detector=None, dtype="fp32"):
        self.c, self.nof_joints, self.checkpoint_path = c, nof_joints, checkpoint_path
        self.model_name, self.resolution = model_name, tuple(resolution)
        self.interpolation = 2 if interpolation is None else int(interpolation)   # cv2.INTER_CUBIC (SimpleHRNet.py:27)
        if self.interpolation not in (0, 1, 2):
            raise ValueError("interpolation: only cv2.INTER_NEAREST (0), cv2.INTER_LINEAR (1) and cv2.INTER_CUBIC (2) are built")
        self.multiperson, self.return_heatmaps = multiperson, return_heatmaps
        self.return_bounding_boxes, self.max_batch_size = return_bounding_boxes, max_batch_size
        if model_name not in ("HRNet", "hrnet", "PoseResNet", "poseresnet", "ResNet", "resnet"):
            raise ValueError("Wrong model name.")                                   # SimpleHRNet.py:114
        if enable_tensorrt:
            raise ValueError("TensorRT is an NVIDIA engine; this class IS the native engine on MI355X")
        self.devices = resolve_devices(device)
        self.device = torch.device("cuda", self.devices[0])      # where results are gathered / single-GPU work runs
        if multiperson and detector is None:
            raise ValueError("multiperson=True needs detector= (the reference's YOLO wrappers are un-vendored third-party code)")
        self.detector = detector
        if len(self.devices) == 1:
            self.model = NativeHRNet(c, nof_joints, self.resolution, dtype, max_batch=max_batch_size, device=self.device,
                                     model_name=model_name)
        else:   # SimpleHRNet.py:126-135: DataParallel over the listed GPUs
            self.model = MultiDeviceHRNet(self.devices, c, nof_joints, self.resolution, dtype, max_batch=max_batch_size,
                                          model_name=model_name)
        if isinstance(checkpoint_path, dict):
            self.model.load_state_dict(checkpoint_path)
        else:
            self.model.load_checkpoint(checkpoint_path)                              # torch.load, raw or {'model': ...}

    # ------------------------------------------------------------------------------------------ SimpleHRNet.py:174-210
    def predict(self, image):
        rank = np.ndim(image)
        if rank == 3:       # one BGR frame
            return self._one_frame(image)
        if rank == 4:       # a stack of frames
            return self._frame_stack(image)
        raise ValueError("Wrong image format.")

    def _result(self, heatmaps, boxes, pts):                                       # :333-343 / :486-496
        wanted = ((self.return_heatmaps, heatmaps), (self.return_bounding_boxes, boxes), (True, pts))
        out = [value for keep, value in wanted if keep]
        return out[0] if len(out) == 1 else out

    def _normalise(self, images_bgr: np.ndarray) -> torch.Tensor:
        """single-person transform (SimpleHRNet.py:213-222 / :355-366): cv2.resize to the model resolution when the frame
        has another size (on the GPU, ``NativeHRNet.resize_frames``), BGR -> RGB, ToTensor, Normalize"""
        if tuple(images_bgr.shape[-3:-1]) != self.resolution:
            return self.model.resize_frames(images_bgr, self.interpolation)
        x = torch.from_numpy(np.ascontiguousarray(images_bgr[..., ::-1])).to(self.device)
        # tensor divisors: torch turns a division by a python scalar into a multiplication by its reciprocal on the GPU,
        # which is not the float32 division ToTensor performs
        x = x.reshape((-1,) + x.shape[-3:]).permute(0, 3, 1, 2).to(torch.float32)
        x = x / torch.full((1, 1, 1, 1), 255.0, dtype=torch.float32, device=self.device)
        mean = torch.tensor(_MEAN, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(_STD, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
        return ((x - mean) / std).contiguous()

    def _hm_shape(self, n):
        return (n, self.nof_joints, self.resolution[0] // 4, self.resolution[1] // 4)

    # ------------------------------------------------------------------------------------------ SimpleHRNet.py:212-343
    def _one_frame(self, image):
        if not self.multiperson:
            images = self._normalise(image)
            boxes = np.asarray([[0, 0, image.shape[1], image.shape[0]]], dtype=np.float32)
            hm, pts = self.model.predict_crops(images, boxes, return_heatmaps=True)
            return self._result(hm.cpu().numpy(), boxes, pts.cpu().numpy())
        found = self.detector.predict_single(image)
        if found is None or len(found) == 0:
            return self._result(np.zeros(self._hm_shape(0), np.float32), np.empty((0, 4), np.int32),
                                np.empty((0, 0, 3), dtype=np.float32))             # :331
        dets = np.asarray(found.cpu() if isinstance(found, torch.Tensor) else found, np.float32)[:, :4]
        out = self.model.predict_frame(image, dets, return_heatmaps=self.return_heatmaps)
        boxes, pts = out[0], out[1].cpu().numpy()
        hm = out[2].cpu().numpy() if self.return_heatmaps else None
        return self._result(hm, boxes, pts)

    # ------------------------------------------------------------------------------------------ SimpleHRNet.py:345-496
    def _frame_stack(self, images):
        if not self.multiperson:
            x = self._normalise(images)
            boxes = np.repeat(np.asarray([[0, 0, images.shape[2], images.shape[1]]], dtype=np.float32), len(images), axis=0)
            hm, pts = self.model.predict_crops(x, boxes, return_heatmaps=True)
            return self._result(hm.cpu().numpy(), boxes, np.expand_dims(pts.cpu().numpy(), axis=1))   # :475
        per_frame = self.detector.predict(images)
        crops, boxes = [], []
        counts = []
        for index, found in enumerate(per_frame):
            n = 0 if found is None else len(found)
            counts.append(None if found is None else n)
            if n:
                dets = np.asarray(found.cpu() if isinstance(found, torch.Tensor) else found, np.float32)[:, :4]
                im, bx, _ = self.model.preprocess_frame(images[index], dets, "clamp")   # :383-412
                crops.append(im), boxes.append(bx)
        if not crops:                                                                # :477-484
            pts = [np.zeros((0, self.nof_joints, 3), dtype=np.float32) for _ in per_frame]
            return self._result(np.zeros(self._hm_shape(0), np.float32), np.asarray([], dtype=np.int32), pts)
        boxes = np.concatenate(boxes, 0)
        out = self.model.predict_crops(torch.cat(crops, 0), boxes, return_heatmaps=self.return_heatmaps)
        hm, pts = (out[0].cpu().numpy(), out[1].cpu().numpy()) if self.return_heatmaps else (None, out.cpu().numpy())
        pts_b, hm_b, boxes_b, index = [], [], [], 0                                  # :445-472: re-add the batch axis
        for n in counts:
            if n is not None:
                pts_b.append(pts[index:index + n])
                if hm is not None:
                    hm_b.append(hm[index:index + n])
                boxes_b.append(boxes[index:index + n])
                index += n
            else:
                pts_b.append(np.zeros((0, self.nof_joints, 3), dtype=np.float32))
                hm_b.append(np.zeros(self._hm_shape(0), dtype=np.float32))
                boxes_b.append(np.zeros((0, 4), dtype=np.float32))
        return self._result(hm_b, boxes_b, pts_b)
