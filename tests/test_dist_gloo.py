"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding / broadcast / all-gather logic of
simple-hrnet_amd/dist.py with a stand-in engine (the real engine needs a GPU; the collective pattern is
identical under RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def test_shard_range_partition():
    sh = load_pkg("dist").shard_range
    for n in (0, 1, 7, 8, 9, 2048, 2049):
        for world in (1, 2, 3, 8):
            r = [sh(n, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))          # contiguous, ordered, disjoint
            assert max(hi - lo for lo, hi in r) <= -(-n // world) if n else True


class FakeNet:
    """deterministic stand-in: 'weights' = blob tensor, pts = f(crops, boxes, blob)"""

    def __init__(self):
        self.blob = torch.zeros(1024, dtype=torch.uint8)
        self.loaded = False

    def load_state_dict(self, sd):
        self.blob.copy_(torch.arange(1024) % 251)
        self.loaded = True

    def weight_blob_tensor(self):
        return self.blob

    def adopt_weights(self):
        self.loaded = True

    def predict_crops(self, images, boxes):
        assert self.loaded
        n = images.shape[0]
        s = images.flatten(1).sum(1, keepdim=True) + float(self.blob.sum())
        b = torch.as_tensor(boxes).float().sum(1, keepdim=True)
        return (s + b).reshape(n, 1, 1).expand(n, 17, 3).contiguous()


def _worker(rank, world, port, n):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = load_pkg("dist")
        eng = sh.ShardedHRNet(FakeNet(), dist)
        eng.load_and_broadcast({"x": 1} if rank == 0 else None, src=0)
        assert eng.net.loaded and int(eng.net.blob.sum()) == int((torch.arange(1024) % 251).sum())
        g = torch.Generator().manual_seed(0)
        images = torch.randn((n, 3, 8, 8), generator=g)
        boxes = torch.randint(0, 100, (n, 4), generator=g, dtype=torch.int32)
        single = FakeNet()
        single.load_state_dict(None)
        want = single.predict_crops(images, boxes)
        got = eng.predict_crops_sharded(images, boxes)          # ragged n: last rank gets a short shard
        assert got.shape == want.shape and torch.equal(got, want)
        lo, hi = sh.shard_range(n - n % world, world, rank)     # equal shards path
        if hi > lo:
            got2 = eng.predict_crops_local_then_gather(images[lo:hi], boxes[lo:hi])
            assert torch.equal(got2, want[: n - n % world])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_world2_gloo(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n), nprocs=2, join=True)
