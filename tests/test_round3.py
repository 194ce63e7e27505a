"""Round-3 odds and ends (VERDICT r2 items 9 / ADVICE r2): PoseResNet-152 on hardware, the block-map cache and the resize tap
table under callers that alternate streams, the launcher detection of the device mapping, hrn_nms's cap / device restore /
scratch release."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_pkg, state_dict_np


# --------------------------------------------------------------------------------------------- device mapping (CPU)
_LAUNCH_VARS = ("LOCAL_RANK", "SLURM_LOCALID", "SLURM_NTASKS", "SLURM_NPROCS", "OMPI_COMM_WORLD_LOCAL_RANK", "OMPI_COMM_WORLD_SIZE",
                "MV2_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_SIZE", "MPI_LOCALRANKID", "PMI_SIZE", "RANK", "WORLD_SIZE")


@pytest.mark.parametrize("env", [{"LOCAL_RANK": "1"}, {"SLURM_LOCALID": "1", "SLURM_PROCID": "5", "SLURM_NTASKS": "8"},
                                 {"OMPI_COMM_WORLD_LOCAL_RANK": "1", "OMPI_COMM_WORLD_SIZE": "2"}])
def test_launchers_other_than_torchrun_get_one_gpu_per_process(monkeypatch, env):
    """srun / mpirun set their own local-rank variables and no LOCAL_RANK: a rank given device='cuda' must take ITS GPU, not
    build an engine on every visible one."""
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert sh.resolve_devices("cuda") == list(range(8))            # a plain process: DataParallel over all of them
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert sh.resolve_devices("cuda") == [1]
    assert sh.resolve_devices("cuda:4,6") == [6]
    assert sh.resolve_devices(torch.device("cuda")) == [1]


def test_srun_with_one_gpu_per_task_and_single_task_jobs(monkeypatch):
    """under `srun --gpus-per-task=1` every task sees ONE GPU at index 0 while SLURM_LOCALID runs 0..7 -- the
    local rank is folded onto the visible devices; and a one-task `srun python ...` is a plain process (all visible GPUs)."""
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("SLURM_LOCALID", "3")
    monkeypatch.setenv("SLURM_NTASKS", "8")
    assert sh.resolve_devices("cuda") == [0] and sh.resolve_device("cuda").index == 0 and sh.resolve_device(None).index == 0
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert sh.resolve_devices("cuda") == [3]
    monkeypatch.setenv("SLURM_NTASKS", "1")                         # single task: not launcher-managed
    monkeypatch.setenv("SLURM_LOCALID", "0")
    assert sh.resolve_devices("cuda") == list(range(8))
    monkeypatch.delenv("SLURM_NTASKS")                              # a local rank but no task count at all: still one GPU
    monkeypatch.setenv("SLURM_LOCALID", "5")                        # per process -- only an EXPLICIT count of 1 makes it a plain process
    assert sh.resolve_devices("cuda") == [5]


def test_local_rank_beyond_the_visible_gpus_is_a_launch_error(monkeypatch):
    """torchrun's LOCAL_RANK is taken as it is (it does not restrict the visible devices), so a rank without a GPU of
    its own is refused with a message that names the cause."""
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert sh.resolve_devices("cuda") == [1]
    monkeypatch.setenv("LOCAL_RANK", "2")
    with pytest.raises(ValueError, match="LOCAL_RANK=2"):
        sh.resolve_devices("cuda")


def test_explicit_device_lists_map_the_local_rank_with_a_modulo(monkeypatch):
    """the range check applies only where LOCAL_RANK IS the device index (None / 'cuda' / torch.device('cuda')).  A launch in
    which every rank sees one GPU (its own CUDA_VISIBLE_DEVICES) and names it 'cuda:0', or two ranks that share one GPU via 'cuda:0,0',
    resolves through ids[local_rank % len(ids)] as before."""
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert sh.resolve_devices("cuda:0") == [0]
    assert sh.resolve_devices("cuda:0,0") == [0]
    assert sh.resolve_devices(torch.device("cuda", 0)) == [0]
    for direct in (None, "cuda", torch.device("cuda")):
        with pytest.raises(ValueError, match="LOCAL_RANK=1"):
            sh.resolve_devices(direct)


def test_rank_and_world_size_without_a_local_rank(monkeypatch):
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setenv("RANK", "6")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert sh.resolve_devices("cuda") == [2]
    monkeypatch.setenv("WORLD_SIZE", "1")                           # a single-process "job": plain-process rules
    assert sh.resolve_devices("cuda") == [0, 1, 2, 3]


def test_torch_device_without_a_visible_gpu_names_gpu_zero(monkeypatch):
    sh = load_pkg("simple_hrnet")
    for k in _LAUNCH_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert sh.resolve_devices(torch.device("cuda")) == [0] and sh.resolve_devices("cuda") == [0]   # (creating the engine then says there is no such device)


def test_nms_refuses_what_its_mask_cannot_hold():
    lib = load_pkg("_lib").load()
    keep = (ctypes.c_int32 * 4)()
    num = ctypes.c_int32()
    boxes = np.zeros((70000, 5), np.float32)
    assert lib.hrn_nms(keep, ctypes.byref(num), boxes.ctypes.data, 70000, 5, ctypes.c_float(0.5), 0) == 2
    assert b"65536" in lib.hrn_nms_last_error()
    assert lib.hrn_nms_release(-1) == 0                              # nothing allocated: a no-op


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_poseresnet152_on_hardware():
    """the largest size the selector offers (models_/poseresnet.py:6-12), never executed before round 3: fp32 against the
    oracle (identical coordinates), bf16 within the usual bound"""
    from oracle import hrnet_torch_oracle as T
    pkg = load_pkg()
    size, h, w, n = 152, 128, 96, 3
    sd = pkg.synth_state_dict(size, 17, 5, model="PoseResNet")
    crops, boxes = pkg.synth_crops(n, h, w, seed=51), pkg.synth_boxes(n, seed=52)
    with torch.no_grad():
        ref = T.poseresnet_forward(pkg.synth.to_torch_state_dict(sd), torch.from_numpy(crops), size).numpy()
    net = pkg.NativeHRNet(size, 17, (h, w), "fp32", max_batch=2, device=0, model_name="PoseResNet").load_state_dict(sd)
    hm, pts = net.predict_crops(torch.from_numpy(crops).cuda(), boxes, return_heatmaps=True)
    assert np.abs(hm.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], T.decode_heatmaps(ref, boxes)[..., :2])
    net.close()
    net = pkg.NativeHRNet(size, 17, (h, w), "bf16", max_batch=4, device=0, model_name="PoseResNet").load_state_dict(sd)
    hb = net(torch.from_numpy(crops).cuda()).cpu().numpy()
    assert np.abs(hb - ref).max() < 0.1 * ref.std() + 0.004
    net.close()


@pytest.mark.gpu
def test_one_handle_under_alternating_streams():
    """block maps are cached per micro-batch size and uploaded on the stream of the call that missed; the resize tap table is
    per handle: calls that alternate streams on one handle (predict_stream, user code under torch.cuda.stream) must neither
    launch ahead of an upload nor rewrite the table under a running kernel"""
    pkg = load_pkg()
    c, h, w = 48, 128, 96
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=8, device=0).load_state_dict(state_dict_np(c))
    x = torch.from_numpy(pkg.synth_crops(8, h, w, seed=3)).cuda()
    want = {n: net(x[:n]).cpu().numpy() for n in (8, 5, 3, 2, 1)}
    net.close()
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=8, device=0).load_state_dict(state_dict_np(c))   # cold caches
    streams = [torch.cuda.Stream() for _ in range(3)]
    frames = [np.random.default_rng(s).integers(0, 256, (2, 90 + 13 * s, 70 + 7 * s, 3), dtype=np.uint8) for s in range(3)]
    ref_frames = [net.resize_frames(f).cpu().numpy() for f in frames]
    torch.cuda.synchronize()
    outs, res = [], []
    for it in range(12):
        n = (8, 5, 3, 2, 1)[it % 5]
        s = streams[it % 3]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append((n, net(x[:n])))
            res.append((it % 3, net.resize_frames(frames[it % 3])))
    torch.cuda.synchronize()
    for n, o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want[n])
    for k, r in res:
        np.testing.assert_array_equal(r.cpu().numpy(), ref_frames[k])
    net.close()


@pytest.mark.gpu
def test_nms_restores_the_callers_device_and_releases_scratch():
    pkg = load_pkg()
    nms = load_pkg("nms")
    lib = load_pkg("_lib").load()
    rng = np.random.default_rng(0)
    d = np.concatenate([rng.uniform(0, 300, (500, 2)), rng.uniform(320, 600, (500, 2)), np.sort(rng.uniform(0, 1, (500, 1)), 0)[::-1]], 1).astype(np.float32)
    torch.cuda.set_device(0)
    keep = nms.gpu_nms(d, 0.5, 0)
    assert torch.cuda.current_device() == 0 and len(keep) > 0
    free0 = torch.cuda.mem_get_info(0)[0]
    assert lib.hrn_nms_release(0) == 0
    assert torch.cuda.mem_get_info(0)[0] >= free0
    np.testing.assert_array_equal(nms.gpu_nms(d, 0.5, 0), keep)     # scratch comes back on demand


def test_release_mode_ignores_the_environment_and_handles_snapshot_their_switches():
    """production ignores every HRN_* switch; a process that sets HRN_DEBUG_ENV=1 opts in and its
    handles snapshot what they saw (hrn_switches) -- checked in fresh processes, the flag being read once per process."""
    import subprocess, sys
    code = ("import importlib, sys; sys.path.insert(0, %r); pkg = importlib.import_module('simple-hrnet_amd');"
            "net = pkg.NativeHRNet(48, 17, (384, 288), 'bf16', max_batch=256, device=-1);"
            "print(repr(net.switches()), sum(i.algo == 3 for i in net.conv_infos()))" % ROOT)
    outs = []
    for env in ({"HRN_DISABLE_N96": "1", "HRN_DEBUG_ENV": "1"}, {"HRN_DISABLE_N96": "1"}, {}):
        e = {k: v for k, v in os.environ.items() if not k.startswith("HRN_")}
        e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120).stdout.strip())
    assert outs[0] == "'HRN_DISABLE_N96=1;' 0"
    assert outs[1] == outs[2] == "'' 144"
