"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

Replaces the reference's only parallel strategy, single-process ``torch.nn.DataParallel``
(SimpleHRNet.py:123-135), which re-broadcasts all 1754 parameter tensors on EVERY forward and gathers
full heat-maps (470 KB/crop) to GPU 0.  Here:
  * weights: ONE broadcast of the packed blob (folded, MFMA-fragment layout) at start-up;
  * crops:   contiguous index ranges, rank r owns [r*ceil(N/G), min(N,(r+1)*ceil(N/G)))  -- every crop is
             independent (eval-mode BatchNorm), so there is no data-path collective inside the network;
  * results: one all-gather of the decoded joints, 17*3*4 = 204 B per crop; heat-maps never leave their GPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    per = -(-n // world) if n > 0 else 0
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class ShardedHRNet:
    """``net``: a ``NativeHRNet`` (or any object with ``predict_crops(images, boxes) -> (n,J,3) tensor``,
    ``load_state_dict``, ``weight_blob_tensor`` and ``adopt_weights``).  ``dist``: the initialised
    ``torch.distributed`` module, or None for a single process."""

    def __init__(self, net, dist=None, group=None):
        self.net, self.dist, self.group = net, dist, group
        self.world = dist.get_world_size(group) if dist else 1
        self.rank = dist.get_rank(group) if dist else 0
        # "nccl" (= RCCL) moves device memory directly.  "gloo" is what the CPU tests use and what several ranks that
        # share ONE GPU have to use (RCCL refuses two ranks on a device): device tensors are staged through the host.
        self.stage_on_host = bool(dist) and dist.get_backend(group) != "nccl"

    def _broadcast(self, t: torch.Tensor, src: int) -> None:
        if t.is_cuda and self.stage_on_host:
            host = t.cpu()
            self.dist.broadcast(host, src=src, group=self.group)
            if self.rank != src:
                t.copy_(host)
                torch.cuda.synchronize(t.device)
        else:
            self.dist.broadcast(t, src=src, group=self.group)

    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        if mine.is_cuda and self.stage_on_host:
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=self.group)
            out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(out, mine.contiguous(), group=self.group)

    # -- weights: rank `src` folds + packs, everybody else receives the blob over RCCL ------------------
    def load_and_broadcast(self, state_dict, src: int = 0) -> None:
        if self.rank == src:
            if state_dict is None:
                raise ValueError("rank %d is the broadcast source and needs the state_dict" % src)
            self.net.load_state_dict(state_dict)
        if self.world > 1:
            blob = self.net.weight_blob_tensor()
            self._broadcast(blob, src)
            if self.rank != src:
                self.net.adopt_weights()

    # -- every rank already holds its own shard (bench / serving loop) ----------------------------------
    def predict_crops_local_then_gather(self, images_local: torch.Tensor, boxes_local) -> torch.Tensor:
        pts = self.net.predict_crops(images_local, boxes_local)
        if self.world == 1:
            return pts
        out = torch.empty((self.world * pts.shape[0],) + tuple(pts.shape[1:]), dtype=pts.dtype, device=pts.device)
        self._all_gather(out, pts)
        return out

    # -- one packed batch known to every rank (the predict() call pattern): shard by index range --------
    def predict_crops_sharded(self, images: torch.Tensor, boxes) -> torch.Tensor:
        n = int(images.shape[0])
        if self.world == 1:
            return self.net.predict_crops(images, boxes)
        lo, hi = shard_range(n, self.world, self.rank)
        per = -(-n // self.world) if n else 0
        b = boxes if isinstance(boxes, torch.Tensor) else torch.as_tensor(boxes)
        pts = self.net.predict_crops(images[lo:hi], b[lo:hi])
        pad = torch.zeros((per,) + tuple(pts.shape[1:]), dtype=pts.dtype, device=pts.device)
        pad[: hi - lo] = pts
        out = torch.empty((self.world * per,) + tuple(pts.shape[1:]), dtype=pts.dtype, device=pts.device)
        self._all_gather(out, pad)
        return out[:n]  # ranges are contiguous and ordered by rank: only the tail is padding
