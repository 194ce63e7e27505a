"""Host logic added in round 2 (no GPU): width validation of the graph compiler, the device-string mapping of the
single-process multi-GPU engine, bench.py's self-describing helpers (source hash gate of the PMC traffic number, the
seeded configs[4] clip, the self-launch command)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_pkg


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


@pytest.mark.parametrize("c", [32, 48, 64, 96])
def test_every_conv_tiles_its_output_channels_exactly(c):
    """a launcher covers cout in tiles of 16*nr channels; a remainder would silently skip channels."""
    pkg = load_pkg()
    for dtype in ("bf16", "fp32"):
        net = pkg.NativeHRNet(c, 17, (128, 96), dtype, max_batch=2, device=-1)
        infos = net.conv_infos()
        assert len(infos) == 291                       # + the stem conv1 and final_layer, which have their own kernels
        for i in infos:
            assert i.nr > 0 and i.cout % (16 * i.nr) == 0, (i.name, i.cout, i.nr)
        net.close()


@pytest.mark.parametrize("c", [16, 80, 112, 40, 0, -32])
def test_unsupported_widths_are_rejected_at_create(c):
    pkg = load_pkg()
    with pytest.raises(ValueError, match="multiple of 32 or of 48"):
        pkg.NativeHRNet(c, 17, (128, 96), "bf16", max_batch=2, device=-1)


def test_device_strings_in_a_plain_process_name_every_listed_gpu(monkeypatch):
    mod = importlib.import_module("simple-hrnet_amd.simple_hrnet")
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert mod.resolve_devices("cuda") == list(range(8))                     # SimpleHRNet.py:128-131: all GPUs
    assert mod.resolve_devices(torch.device("cuda")) == list(range(8))
    assert mod.resolve_devices("cuda:1,2") == [1, 2] and mod.resolve_devices("cuda:0,0") == [0, 0]   # :132-135
    assert mod.resolve_devices("cuda:3") == [3] and mod.resolve_devices(torch.device("cuda", 5)) == [5]
    assert mod.resolve_devices(None) == [0]
    for bad in ("cuda:1,x", "cuda:-1,2", "cuda:"):
        with pytest.raises(ValueError, match="Wrong device name."):
            mod.resolve_devices(bad)
    with pytest.raises(ValueError, match="no CPU path"):
        mod.resolve_devices("cpu")
    monkeypatch.setenv("LOCAL_RANK", "1")                                    # under torch.distributed.run: one GPU per process
    assert mod.resolve_devices("cuda") == [1] and mod.resolve_devices("cuda:4,6") == [6] and mod.resolve_devices("cuda:5") == [5]


def test_multi_device_engine_shards_like_the_ranks_do(monkeypatch):
    """MultiDeviceHRNet's index ranges, threading and gather on stand-in engines (no GPU): same rows, same order."""
    native = importlib.import_module("simple-hrnet_amd.native")

    class Fake:
        def __init__(self, c, j, res, dtype, max_batch, device, model_name):
            self.c, self.nof_joints, self.resolution, self.dtype, self.max_batch, self.model_name = c, j, res, dtype, max_batch, model_name
            self.device_index, self.calls = device, []

        torch_device = torch.device("cpu")

        def predict_crops(self, images, boxes, return_heatmaps=False):
            self.calls.append(int(images.shape[0]))
            pts = images.reshape(images.shape[0], -1)[:, :1].reshape(-1, 1, 1).expand(-1, 17, 3).contiguous()
            return (torch.zeros((images.shape[0], 17, 2, 2)), pts) if return_heatmaps else pts

        def close(self):
            pass

    class NoStream:
        def wait_event(self, e):
            pass

    class NoEvent:
        def record(self, s=None):
            pass

    monkeypatch.setattr(native, "NativeHRNet", Fake)
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: NoStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: NoEvent())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: NoStream())
    import contextlib
    monkeypatch.setattr(torch.cuda, "device", lambda *a, **k: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda *a, **k: contextlib.nullcontext())
    eng = native.MultiDeviceHRNet([0, 1, 2], 32, 17, (64, 48), "fp32", max_batch=4)
    x = torch.arange(7, dtype=torch.float32).reshape(7, 1, 1, 1).expand(7, 3, 64, 48).contiguous()
    pts = eng.predict_crops(x, np.zeros((7, 4), np.int32))
    assert pts[:, 0, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert [n.calls for n in eng.nets] == [[3], [3], [1]]                    # ceil(7/3) = 3 per engine, the last one short
    hm, pts = eng.predict_crops(x[:2], np.zeros((2, 4), np.int32), return_heatmaps=True)
    assert tuple(hm.shape) == (2, 17, 2, 2) and pts[:, 0, 0].tolist() == [0, 1]
    assert [n.calls for n in eng.nets] == [[3, 1], [3, 1], [1]]              # 2 crops: one each on the first two engines
    eng.close()


def test_bench_source_hash_gates_the_pmc_traffic_number(tmp_path, monkeypatch):
    b = _bench()
    h = b.source_hash()
    assert len(h) == 16 and h == b.source_hash()

    class A:
        c, dtype, height, width = 48, "bf16", 384, 288

    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "simple-hrnet_amd" / "csrc")
    (tmp_path / "simple-hrnet_amd" / "csrc" / "k.hip").write_text("// v1\n")
    cur = b.source_hash()
    assert b.pmc_traffic(A, 256) == (None, "no PMC reading committed")
    (tmp_path / "profiles" / "round2_pmc_traffic.json").write_text(json.dumps({"traffic_bytes_per_launch": 123, "source_hash": cur}))
    t, src = b.pmc_traffic(A, 256)
    assert t == 123 and "round2_pmc_traffic.json" in src
    (tmp_path / "simple-hrnet_amd" / "csrc" / "k.hip").write_text("// v2: the kernel changed\n")
    t, src = b.pmc_traffic(A, 256)
    assert t is None and src.startswith("stale")                             # never quote a counter reading of other code
    assert b.pmc_traffic(A, 128)[0] is None                                   # nor one of another workload


def test_bench_clip_is_seeded_and_in_detector_format():
    b = _bench()
    clip, dets = b.make_clip(frames=3, hf=270, wf=480, people=4)
    clip2, dets2 = b.make_clip(frames=3, hf=270, wf=480, people=4)
    assert clip.shape == (3, 270, 480, 3) and clip.dtype == np.uint8 and np.array_equal(clip, clip2) and np.array_equal(dets, dets2)
    assert dets.shape == (3, 4, 7) and dets.dtype == np.float32             # x1, y1, x2, y2, conf, cls_conf, cls_pred
    _, dets = b.make_clip(frames=2)
    assert dets.shape == (2, 8, 7)
    assert (dets[..., 0] >= 0).all() and (dets[..., 2] < 1920).all() and (dets[..., 3] < 1080).all()
    assert (dets[..., 2] > dets[..., 0]).all() and (dets[..., 3] > dets[..., 1]).all()


def test_bench_spawns_its_own_ranks_when_not_launched(monkeypatch):
    """`python bench.py --gpus N` (the form the driver uses) must not ask the user for a launcher: it re-executes itself
    under torch.distributed.run with one rank per GPU on 127.0.0.1 and relays rank 0's JSON line."""
    b = _bench()
    seen = {}

    class R:
        returncode = 0
        stdout = 'noise\n{"n_gpus": 4}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        return R()

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    b.main()
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        b.main()
