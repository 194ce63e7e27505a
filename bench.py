#!/usr/bin/env python3
"""Benchmark of the HRNet hot path (model call + decode, SimpleHRNet.py:281-308) on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): person-crops/sec, HRNet-W48 384x288, batch 256 per GPU, bf16 MFMA with fp32
accumulate, synthetic crops already resident in HBM, random-init (seeded) weights.  One "step" = one pass
of the hot path over one batch of 256 crops per GPU: stem -> stage1-4 -> head -> arg-max decode ->
(N>1) all-gather of the keypoints.  N>1: one process per GPU, RCCL backend; crops are sharded by contiguous
index ranges (weak scaling: 256 crops per GPU), the packed weights are broadcast once from rank 0, and the only
per-step collective is the all-gather of 204 B/crop of joints.  `python bench.py --gpus N` spawns its own N
ranks (re-executes itself under torch.distributed.run) when it was not started by a launcher; started BY a
launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.

Prints ONE JSON line on rank 0: the driver-contract fields plus
  roofline      dominant kernel (stage-3/4 BasicBlock 3x3 convs) against the dense bf16 MFMA peak, HIP-event timed
  parity        N=1: the first 32 crops of the TIMED batch against the CPU oracle (fp32 engine: identical coordinates;
                bf16 engine: arg-max agreement and pixel histogram)
  cpu_baseline  N=1: the reference's device='cpu' path restated (oracle), max_batch_size=32 chunks, on this box's cores
  clip          N=1: BASELINE configs[4] -- 30 synthetic 1080p frames x 8 detector boxes, frame -> crops -> joints
  parity.peaked N=1: the timed configuration on the PEAKED checkpoint (synth.peaked_state_dict) against the fp32 CPU oracle and
                against the cell the construction puts every joint's peak in
  config1_fp32  N=1: BASELINE configs[1] -- HRNet-W32 256x192, batch 64, fp32, against the fp32 MFMA peak
  fp32_w48_384x288  N=1: the headline shape in the parity mode (fp32), timed
  prepath, pcie_inclusive   side measurements
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_F32_TFLOPS = 157.3     # fp32 MFMA = vector rate
PARITY_CROPS = 32           # = one max_batch_size chunk of the reference (SimpleHRNet.py:285-294)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="crops per GPU per step")
    ap.add_argument("--c", type=int, default=48, help="HRNet width, or the ResNet size with --model-name PoseResNet")
    ap.add_argument("--model-name", default="HRNet", choices=["HRNet", "PoseResNet"],
                    help="side measurements only: the headline metric is HRNet-W48 (BASELINE.json)")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=288)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--max-batch", type=int, default=int(os.environ.get("HRN_MAX_BATCH", "256")),
                    help="crops per internal pass (workspace size)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline AND parity (both need the CPU oracle)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline sample")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-prepath", action="store_true", help="skip the crop pre-path / PCIe side measurements")
    ap.add_argument("--no-clip", action="store_true", help="skip the configs[4] clip block")
    ap.add_argument("--no-config1", action="store_true", help="skip the configs[1] fp32 side line")
    ap.add_argument("--no-peaked", action="store_true", help="skip parity.peaked (bf16 px-match on the peaked checkpoint)")
    ap.add_argument("--no-two-lanes", action="store_true", help="skip the side measurement with two engines sharing the GPU")
    ap.add_argument("--no-fp32-w48", action="store_true", help="skip the fp32 line on the headline shape")
    ap.add_argument("--clip", action="store_true", help="ONLY the configs[4] clip measurement (its JSON line is the clip block)")
    ap.add_argument("--check-gather", action="store_true", help="N > 1: compare the all-gathered joints with one engine run over every rank's crops")
    ap.add_argument("--zeros", action="store_true",
                    help="DIAGNOSTIC (never the metric): all-zero weights, BN shifts and crops -- every MFMA operand is zero, so the part's "
                         "power budget is out of the picture; against the random run on the same box it tells a power-bound kernel (>= 15 %% faster) "
                         "from a stall-bound one (VERDICT r5 item 1a)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# CPU legs (child process): the oracle restates the reference's device='cpu' path; it is the CHECKER and the reported
# CPU baseline, never part of what is timed on the GPU.
def _host_cpu():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    logical = os.cpu_count() or 1
    quota = logical
    try:  # container CPU quota (cgroup v2): threads beyond it only thrash
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, min(logical, int(int(q) / int(period))))
    except Exception:
        pass
    return model, logical, quota


def _cpu_worker(spec_path):
    """child process body: prints one JSON line.  spec: {c,h,w,budget_s, crops (npy path or None), boxes, out (npz path),
    clip: {...} or None}"""
    import numpy as np
    import torch

    spec = json.load(open(spec_path))
    pkg = importlib.import_module("simple-hrnet_amd")
    from oracle import hrnet_torch_oracle as T

    c, h, w, budget_s = spec["c"], spec["h"], spec["w"], spec["budget_s"]
    sd = pkg.synth.to_torch_state_dict(pkg.synth_state_dict(c, 17, 0))
    chunk = PARITY_CROPS
    if spec.get("crops"):
        crops = torch.from_numpy(np.load(spec["crops"]))
        boxes = np.asarray(spec["boxes"], np.int32)
    else:
        crops = torch.from_numpy(pkg.synth_crops(chunk, h, w))
        boxes = pkg.synth_boxes(chunk)
    model, logical, quota = _host_cpu()
    # one thread per logical CPU collapses oneDNN on big SMT hosts; probe a few thread counts on two crops
    # and keep the fastest (the count actually used is reported as `cores`)
    best, best_t = None, 1e30
    for nt in sorted({min(quota, x) for x in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        T.predict_crops(sd, crops[:1], boxes[:1])          # warm-up (primitive cache)
        t0 = time.perf_counter()
        T.predict_crops(sd, crops[:2], boxes[:2])
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
        if dt > budget_s / 2:
            break
    torch.set_num_threads(best)
    # the sample: whole max_batch_size = 32 chunks (the reference's chunk loop, SimpleHRNet.py:285-294) of the same
    # workload until the budget is spent; the first chunk's outputs are the parity reference
    done, t0, first = 0, time.perf_counter(), None
    while True:
        hm, pts = T.predict_crops(sd, crops, boxes, max_batch_size=chunk)
        if first is None:
            first = (hm, pts)
        done += len(crops)
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 512:
            break
    if spec.get("out"):
        extra = {}
        pk = spec.get("peaked")
        if pk:
            # the peaked checkpoint (synth.peaked_state_dict: random trunk + three signal channels on the identity paths) on a few
            # crops spread through the batch the GPU runs, on-cell and off-cell blob centres; outside the timed sample
            psd = pkg.synth.to_torch_state_dict(pkg.synth.peaked_state_dict(c, 17, 0))
            for tag, on_cell in (("on", True), ("off", False)):
                pc, _ = pkg.synth.peaked_crops(pk["n"], h, w, seed=pk["seed"], on_cell=on_cell)
                extra["peaked_%s_hm" % tag] = T.hrnet_forward(psd, torch.from_numpy(pc[pk["idx"]])).numpy()
        if spec.get("emulate_bf16"):
            # the engine-arithmetic restatement (bf16 weights and stored activations, fp32 accumulation) of the same crops:
            # what the bf16 kernels are pinned to; outside the timed sample
            extra["emu_hm"] = T.hrnet_forward_engine(sd, crops[:chunk]).numpy()
        np.savez(spec["out"], hm=first[0], pts=first[1], **extra)
    res = {"value": round(done / el, 3), "unit": "crops/s", "cores": best, "kind": "port", "cpu": model,
           "sample": "%d crops (max_batch_size=%d chunks, SimpleHRNet.py:285-294) of HRNet-W%d %dx%d fp32 incl. decode, "
                     "torch-CPU restatement of the reference device='cpu' path (oracle/hrnet_torch_oracle.py; /root/reference "
                     "does not exist on the GPU box), %d threads (container quota %d CPUs of %d logical), %.1f s"
                     % (done, chunk, c, h, w, best, quota, logical, el)}
    clip = spec.get("clip")
    if clip:  # the reference's per-frame loop (SimpleHRNet.py:228-308): PIL crop/pad/resize per person, model, decode
        from oracle import prepath_oracle as P

        frame = np.load(clip["frame"])
        dets = np.asarray(clip["dets"], np.float32)
        t0 = time.perf_counter()
        images, bx = P.prepath(frame, dets, h, w, resize=P.pil_resize)
        _, cpts = T.predict_crops(sd, torch.from_numpy(images), bx, max_batch_size=chunk)
        el = time.perf_counter() - t0
        np.savez(clip["out"], pts=cpts, boxes=bx)
        res["clip"] = {"fps": round(1.0 / el, 4), "persons_per_s": round(len(dets) / el, 3), "frames": 1,
                       "sample": "1 frame x %d people through the PIL pre-path + model + decode, %d threads" % (len(dets), best)}
    print(json.dumps(res))


def _run_cpu_worker(spec, budget_s):
    import subprocess

    fd, path = tempfile.mkstemp(suffix=".json", prefix="hrn_cpu_")
    with os.fdopen(fd, "w") as f:
        json.dump(spec, f)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", path]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 8 + 120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout or failure: report it, never block the GPU number
        return {"value": None, "unit": "crops/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "cpu baseline did not finish: %s" % type(e).__name__}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


# ----------------------------------------------------------------------------------------------------------------------
def source_hash():
    """hash of the kernel / host sources a counter reading belongs to (the GPU box has no .git)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "simple-hrnet_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(a, nb):
    """HBM-side bytes per grouped launch from the committed PMC passes (counters cannot be read from inside the process).
    Only a reading taken on EXACTLY these sources is quoted; anything else is reported as null with the reason."""
    import glob

    if not (a.c == 48 and a.dtype == "bf16" and nb == 256 and (a.height, a.width) == (384, 288)):
        return None, "no PMC reading for this configuration"
    cur = source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("source_hash") == cur:
            return j["traffic_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, sources %s)" % (os.path.basename(path), cur)
        stale = stale or os.path.basename(path)
    return None, "stale: %s was taken on other sources than %s -- re-run tools/pmc_traffic.sh" % (stale, cur) if stale else "no PMC reading committed"


def make_clip(seed=7, frames=30, people=8, hf=1080, wf=1920):
    """BASELINE configs[4]: a seeded synthetic 1080p clip and, per frame, `people` boxes in the detector's (P, 7) format
    (x1, y1, x2, y2, conf, cls_conf, cls_pred -- models_/detectors/YOLOv3.py:79-141), standing upright, inside the frame."""
    import numpy as np

    rng = np.random.default_rng(seed)
    clip = rng.integers(0, 256, (frames, hf, wf, 3), dtype=np.uint8)
    dets = np.zeros((frames, people, 7), np.float32)
    for f in range(frames):
        for i in range(people):
            bh = rng.integers(hf * 5 // 18, hf * 5 // 6)           # 300..900 px tall on a 1080p frame
            bw = int(bh * rng.uniform(0.3, 0.6))
            x1, y1 = rng.uniform(0, wf - bw - 1), rng.uniform(0, hf - bh - 1)
            dets[f, i] = (x1, y1, x1 + bw, y1 + bh, 0.9, 0.9, 0.0)
    return clip, dets


def run_clip(net, clip_host, dets, mode="per_frame", rank=0, world=1):
    """frame (uint8, pinned host) -> HBM once -> hrn_preprocess_frame -> hrn_forward -> joints.  `per_frame`: the live
    loop (scripts/live-demo.py:93-162 minus the camera and the detector network), frames dealt round-robin to the ranks,
    everything asynchronous on one stream, ONE read-back of all joints at the end; `per_frame_sync`: the same with the
    204 B/person read back after every frame (what a display loop needs); `stacked`: SimpleHRNet.predict on the 4-D
    stack -- every frame's crops packed into one batch (SimpleHRNet.py:345-443).  Returns (pts (F,P,J,3) numpy with
    zeros for other ranks' frames, seconds)."""
    import numpy as np
    import torch

    dev = net.torch_device
    nf, people = dets.shape[0], dets.shape[1]
    mine = [f for f in range(nf) if f % world == rank]
    out = torch.zeros((nf, people, net.nof_joints, 3), dtype=torch.float32, device=dev)
    host = np.zeros((nf, people, net.nof_joints, 3), np.float32)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if mode == "stacked":
        crops, boxes = [], []
        for f in mine:
            im, _, bdev = net.preprocess_frame(clip_host[f].to(dev, non_blocking=True), dets[f], "clamp")
            crops.append(im), boxes.append(bdev)
        if crops:
            pts = net.predict_crops(torch.cat(crops, 0), torch.cat(boxes, 0))
            out[mine] = pts.view(len(mine), people, net.nof_joints, 3)
        host = out.cpu().numpy()
    else:
        for f in mine:
            _, pts = net.predict_frame(clip_host[f].to(dev, non_blocking=True), dets[f])
            if mode == "per_frame_sync":
                host[f] = pts.cpu().numpy()
            else:
                out[f] = pts
        if mode != "per_frame_sync":
            host = out.cpu().numpy()
    torch.cuda.synchronize(dev)
    return host, time.perf_counter() - t0


def run_clip_pipelined(nets, streams, clip_host, dets, rank=0, world=1):
    """`per_frame` with several frames in flight: this rank's frames are dealt round-robin to L engines (handles) of the same
    GPU, each on its own stream -- the pass of one frame (8 crops: ~135 small launches, most of them a fraction of the chip)
    runs beside the passes of the next ones.  Latency per frame is that of `per_frame`; throughput is what a serving loop
    with a small queue sees.  One host thread issues everything; ONE read-back at the end."""
    import numpy as np
    import torch

    dev = nets[0].torch_device
    nf, people = dets.shape[0], dets.shape[1]
    mine = [f for f in range(nf) if f % world == rank]
    out = torch.zeros((nf, people, nets[0].nof_joints, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    main = torch.cuda.current_stream(dev)
    for st in streams:
        st.wait_stream(main)
    for k, f in enumerate(mine):
        lane = k % len(nets)
        with torch.cuda.stream(streams[lane]):
            _, pts = nets[lane].predict_frame(clip_host[f].to(dev, non_blocking=True), dets[f])
            out[f] = pts
    for st in streams:
        main.wait_stream(st)
    host = out.cpu().numpy()
    torch.cuda.synchronize(dev)
    return host, time.perf_counter() - t0


def clip_measure(pkg, net, dist, rank, world, cpu_clip=None):
    import numpy as np
    import torch

    clip, dets = make_clip()
    clip_host = torch.from_numpy(clip).pin_memory()
    res = {"workload": "BASELINE configs[4]: %d synthetic %dx%d uint8 frames, %d boxes per frame in the detector's (P,7) format; "
                       "per frame: upload once -> crop/pad/resize/normalise on the GPU -> HRNet-W%d %dx%d %s -> decode -> 204 B/person "
                       "back (the detector network itself is out of scope, SURVEY.md 8)" %
                       (clip.shape[0], clip.shape[1], clip.shape[2], dets.shape[1], net.c, net.resolution[0], net.resolution[1], net.dtype),
           "n_gpus": world, "frames": int(clip.shape[0]), "persons": int(dets.shape[0] * dets.shape[1])}
    ref = None
    for mode in ("per_frame", "per_frame_sync", "stacked"):
        run_clip(net, clip_host, dets, mode, rank, world)                      # warm-up (allocator, block maps)
        if dist:
            dist.barrier()
        pts, el = run_clip(net, clip_host, dets, mode, rank, world)
        if dist:
            t = torch.tensor([el], device=net.torch_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        res[mode] = {"fps": round(clip.shape[0] / el, 2), "persons_per_s": round(dets.shape[0] * dets.shape[1] / el, 1),
                     "ms_per_frame": round(el / clip.shape[0] * 1e3, 3)}
        if mode == "per_frame":
            ref = pts
        elif mode == "per_frame_sync" and ref is not None:
            res["per_frame_sync"]["same_joints_as_per_frame"] = bool(np.array_equal(pts, ref))
    # several frames in flight: L small engines (max_batch = people per frame) sharing the GPU, weights copied blob to blob.
    # A single-GPU side measurement (no collectives inside the try: a rank that failed here must not leave the others at a barrier)
    if world == 1:
        try:
            lanes = int(os.environ.get("HRN_CLIP_LANES", "3"))
            smalls = [pkg.NativeHRNet(net.c, net.nof_joints, net.resolution, net.dtype, max_batch=int(dets.shape[1]), device=net.device_index,
                                      model_name=net.model_name) for _ in range(lanes)]
            for sm in smalls:
                sm.weight_blob_tensor().copy_(net.weight_blob_tensor())
                torch.cuda.synchronize(net.torch_device)
                sm.adopt_weights()
            streams = [torch.cuda.Stream(net.torch_device) for _ in range(lanes)]
            run_clip_pipelined(smalls, streams, clip_host, dets)
            pts, el = run_clip_pipelined(smalls, streams, clip_host, dets)
            res["per_frame_pipelined"] = {"fps": round(clip.shape[0] / el, 2), "persons_per_s": round(dets.shape[0] * dets.shape[1] / el, 1),
                                          "ms_per_frame": round(el / clip.shape[0] * 1e3, 3), "frames_in_flight": lanes,
                                          "same_joints_as_per_frame": bool(ref is not None and np.array_equal(pts, ref)),
                                          "workspace_bytes_per_engine": int(smalls[0].workspace_bytes()),
                                          "workspace_bytes_total": int(sum(sm.workspace_bytes() for sm in smalls)),
                                          "weight_blob_bytes_per_engine": int(smalls[0].weight_blob_tensor().numel()),
                                          "note": "throughput with %d frames in flight on %d engines of one GPU; the latency of a frame is per_frame's" % (lanes, lanes)}
            for sm in smalls:
                sm.close()
        except Exception as e:   # a side measurement: never lose the main ones to it
            res["per_frame_pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
    res["fps"] = res["per_frame"]["fps"]
    if cpu_clip and "fps" in cpu_clip and rank == 0:
        res["cpu_reference_loop"] = {k: cpu_clip[k] for k in ("fps", "persons_per_s", "frames", "sample")}
        res["speedup_vs_cpu_loop"] = round(res["fps"] / cpu_clip["fps"], 1)
    return res, (clip, dets, ref)


def prepath_measure(pkg, net, dev):
    """Side measurement (not part of `value`): the crop pre-path of SimpleHRNet.py:236-278 on a synthetic 1080p frame
    with 16 people: GPU kernel, frame resident in HBM."""
    import numpy as np
    import torch

    rng = np.random.default_rng(5)
    hf, wf, people = 1080, 1920, 16
    frame = rng.integers(0, 256, (hf, wf, 3), dtype=np.uint8)
    dets = np.zeros((people, 4), np.float32)
    for i in range(people):
        bh = rng.integers(300, 900)
        bw = int(bh * rng.uniform(0.3, 0.6))
        x1, y1 = rng.uniform(0, wf - bw), rng.uniform(0, hf - bh)
        dets[i] = (x1, y1, x1 + bw, y1 + bh)
    fdev = torch.from_numpy(frame).to(dev)
    for _ in range(3):
        net.preprocess_frame(fdev, dets)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        net.preprocess_frame(fdev, dets)
    torch.cuda.synchronize()
    gpu = reps * people / (time.perf_counter() - t0)
    h, w = net.resolution
    return {"crops_per_s": round(gpu, 1), "unit": "crops/s",
            "workload": "1920x1080 uint8 frame resident in HBM, 16 people, crop+pad+resize+normalise to %dx%d" % (h, w)}


def pcie_measure(pkg, net, batch, dev):
    """Side measurement (never `value`): the same pass fed from pinned HOST memory through NativeHRNet.predict_stream --
    batch k+1 crosses PCIe on a copy stream while batch k computes -- and, for comparison, upload-then-compute in series."""
    import torch

    h, w = net.resolution
    host = [torch.randn((batch, 3, h, w), dtype=torch.float32).pin_memory() for _ in range(2)]
    boxes = pkg.synth_boxes(batch)
    reps = 6

    def overlapped():
        for _ in net.predict_stream(((host[k & 1], boxes) for k in range(reps))):
            pass
        torch.cuda.synchronize()

    def serial():
        for k in range(reps):
            net.predict_crops(host[k & 1].to(dev, non_blocking=True), boxes)
        torch.cuda.synchronize()

    # the same crops as uint8 BGR at the network's resolution (what cv2.resize leaves on the host): a quarter of the bytes, the colour
    # flip + ToTensor + Normalize on the GPU
    host8 = [torch.randint(0, 256, (batch, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]

    def overlapped_u8():
        for _ in net.predict_stream(((host8[k & 1], boxes) for k in range(reps))):
            pass
        torch.cuda.synchronize()

    def serial_u8():
        for k in range(reps):
            net.predict_crops(net.resize_frames(host8[k & 1], 0), boxes)
        torch.cuda.synchronize()

    # what the copy engine and the compute do to each other, measured apart: one upload alone, one pass alone, both at once
    dev_buf = torch.empty((batch, 3, h, w), dtype=torch.float32, device=dev)
    resident = torch.randn((batch, 3, h, w), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(dev)

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def both():
        with torch.cuda.stream(side):
            dev_buf.copy_(host[0], non_blocking=True)
        net.predict_crops(resident, boxes)
        torch.cuda.current_stream(dev).wait_stream(side)
    t_copy = timed(lambda: dev_buf.copy_(host[0], non_blocking=True))
    t_pass = timed(lambda: net.predict_crops(resident, boxes))
    t_both = timed(both)

    res = {}
    for name, fn in (("overlapped", overlapped), ("serial", serial), ("u8_overlapped", overlapped_u8), ("u8_serial", serial_u8)):
        fn()
        t0 = time.perf_counter()
        fn()
        res[name] = reps * batch / (time.perf_counter() - t0)
    return {"crops_per_s": round(res["overlapped"], 1), "unit": "crops/s", "serial_crops_per_s": round(res["serial"], 1),
            "uint8_crops_per_s": round(res["u8_overlapped"], 1), "uint8_serial_crops_per_s": round(res["u8_serial"], 1),
            "apart_ms": {"upload_alone": round(t_copy, 3), "pass_alone": round(t_pass, 3), "both_at_once": round(t_both, 3),
                         "upload_GBps": round(batch * 3 * h * w * 4 / t_copy / 1e6, 1),
                         "note": "both_at_once ~ max(...) = the copy engine works beside the kernels; ~ sum(...) = the upload is a shader "
                                 "blit that waits for CUs the persistent kernels hold"},
            "workload": "%d batches of %d crops in pinned host memory, uploads on a copy stream behind the previous batch's compute: "
                        "fp32 NCHW (%.0f MB per batch) and uint8 NHWC BGR (%.0f MB, colour flip + normalisation on the GPU)"
                        % (reps, batch, batch * 3 * h * w * 4 / 1e6, batch * 3 * h * w / 1e6)}


def config1_measure(pkg, dev):
    """BASELINE configs[1]: HRNet-W32 256x192, batch 64 random crops, one GPU, fp32 (exact-fp32 MFMA) -- the parity mode."""
    import torch

    c, h, w, n = 32, 256, 192, 64
    net = pkg.NativeHRNet(c, 17, (h, w), "fp32", max_batch=n, device=dev.index).load_state_dict(pkg.synth_state_dict(c, 17, 0))
    g = torch.Generator(device=dev).manual_seed(77)
    images = torch.randn((n, 3, h, w), generator=g, device=dev, dtype=torch.float32)
    boxes = torch.from_numpy(pkg.synth_boxes(n, seed=77)).to(dev)
    for _ in range(2):
        net.predict_crops(images, boxes)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        net.predict_crops(images, boxes)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    value = steps * n / el
    flops = net.flops_per_crop()
    conv_ms, other = net.profile_pass(images)
    conv_ms, other = net.profile_pass(images)
    tf = value * flops / 1e12
    infos = net.conv_infos()
    conv_tf = sum(i.flops for i in infos) * n / (sum(conv_ms) * 1e-3) / 1e12
    lds = [k for k, i in enumerate(infos) if i.algo == 1]           # the LDS-staged fp32 kernel (conv3x3_f32.hip): the 3x3 stride-1 convs
    lds_tf = sum(infos[k].flops for k in lds) * n / (sum(conv_ms[k] for k in lds) * 1e-3) / 1e12 if lds else 0.0
    net.close()
    return {"workload": "HRNet-W32 256x192, batch=64 random crops, fp32 (v_mfma_f32_16x16x4_f32), model forward + decode (BASELINE configs[1])",
            "value": round(value, 1), "unit": "crops/s", "ms_per_step": round(el / steps * 1e3, 3), "dtype": "f32",
            "gflop_per_crop": round(flops / 1e9, 3), "whole_net_tflops": round(tf, 2),
            "roofline": {"bound": "mfma", "kernel": "all convolutions of the pass, fp32: the 3x3 stride-1 ones on conv3x3_f32_kernel (LDS-staged, "
                                                      "%d convs, %.1f TFLOP/s = %.3f of the peak on their own), the rest on the generic MFMA kernel" % (len(lds), lds_tf, lds_tf / PEAK_F32_TFLOPS),
                         "achieved": round(conv_tf, 2),
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(conv_tf / PEAK_F32_TFLOPS, 4),
                         "whole_net_frac": round(tf / PEAK_F32_TFLOPS, 4),
                         "timing": "HIP events on the launch stream around every kernel of one pass of 64 crops"}}


PEAKED_IDX = [0, 37, 74, 111, 148, 185, 222, 255]   # crops of the 256-crop peaked batch the CPU oracle also evaluates
PEAKED_SEED = 5


def peaked_measure(pkg, a, dev, cpu_out):
    """bf16 px-match on heat-maps that HAVE a peak: the engine in the timed configuration (same dtype, batch,
    micro-batch) on the peaked checkpoint / crops of synth.py, against (i) the fp32 CPU oracle on PEAKED_IDX and (ii) the cell the
    construction puts the peak in, for every (crop, joint) of the batch."""
    import numpy as np
    import torch

    ref = np.load(cpu_out)
    n, h4, w4 = a.batch, a.height // 4, a.width // 4
    idx = [i for i in PEAKED_IDX if i < n]
    net = pkg.NativeHRNet(a.c, 17, (a.height, a.width), a.dtype, max_batch=a.max_batch, device=dev.index)
    sdp = pkg.synth.peaked_state_dict(a.c, 17, 0)
    net.load_state_dict(sdp)
    mix = sdp["final_layer.weight"][:, :pkg.synth.PEAKED_SIGNALS, 0, 0]
    boxes = torch.from_numpy(np.tile(np.asarray([[0, 0, a.width, a.height]], np.int32), (n, 1))).to(dev)   # box = the crop: pts in crop px
    out = {"weights": "synth.peaked_state_dict(c, 17, 0): the seeded random checkpoint with three signal channels on the identity paths "
                      "(leak 0.05 from the random channels at every layer), head = signal mix + 0.25 x random",
           "crops": "synth.peaked_crops(%d, %d, %d, seed %d): one blob (sigma 6 px, amplitude 20-30) per colour plane on N(0, 0.2) noise" % (n, a.height, a.width, PEAKED_SEED),
           "reference": "fp32 CPU oracle on crops %s of the batch; construction (strongest weighted blob's cell) on all %d x 17 joints" % (idx, n),
           "px_unit": "crop pixels (one heat-map cell = 4 px; the reference decodes without sub-cell refinement, SimpleHRNet.py:297-308)"}
    for tag, on_cell in (("on", True), ("off", False)):
        key = "peaked_%s_hm" % tag
        if key not in ref.files:
            continue
        crops, cen = pkg.synth.peaked_crops(n, a.height, a.width, seed=PEAKED_SEED, on_cell=on_cell)
        hm, pts = net.predict_crops(torch.from_numpy(crops).to(dev), boxes, return_heatmaps=True)
        hm, pts = hm.cpu().numpy(), pts.cpu().numpy()
        ref_hm = ref[key]
        k = len(idx)
        flat, rflat = hm[idx].reshape(k, 17, -1), ref_hm.reshape(k, 17, -1)
        am, ram = flat.argmax(-1), rflat.argmax(-1)
        cells = np.maximum(np.abs(am // w4 - ram // w4), np.abs(am % w4 - ram % w4))
        srt = np.sort(rflat, -1)
        margin = srt[..., -1] - srt[..., -2]
        err = float(np.abs(hm[idx] - ref_hm).max())
        safe = margin > 4 * err
        # construction: joint j peaks on the blob s maximising mix[j, s] * amplitude -- take the blob nearest to the engine's peak
        amall = hm.reshape(n, 17, -1).argmax(-1)
        py, px = amall // w4, amall % w4
        d = np.min(np.maximum(np.abs(py[:, :, None] - cen[:, None, :, 0] / 4.0), np.abs(px[:, :, None] - cen[:, None, :, 1] / 4.0)), -1)
        res = {"argmax_agree_frac": round(float((am == ram).mean()), 4), "joints_compared": int(am.size),
               "max_dev_px": int(cells.max() * 4), "px_hist": {"0": int((cells == 0).sum()), "4": int((cells == 1).sum()), ">4": int((cells > 1).sum())},
               "max_abs_dH": round(err, 4), "peak_median": round(float(np.median(srt[..., -1])), 3),
               "margin_min": round(float(margin.min()), 4), "margin_median": round(float(np.median(margin)), 4),
               "frac_margin_gt_4err": round(float(safe.mean()), 4),
               "agree_where_margin_gt_4err": (round(float((am == ram)[safe].mean()), 4) if safe.any() else None),
               "all_joints_max_cells_from_a_blob_centre": round(float(d.max()), 3),
               "all_joints_on_a_blob_cell_frac": round(float((d <= (0.0 if on_cell else 0.75)).mean()), 4)}
        out["on_cell" if on_cell else "off_cell"] = res
    net.close()
    return out


def fp32_w48_measure(pkg, a, dev):
    """The configuration in which +-0.5 px holds today -- fp32 (exact-fp32 MFMA) on the HEADLINE shape: HRNet-W48 384x288, batch
    256 -- timed like the headline."""
    import torch

    c, h, w, n = 48, 384, 288, 256
    net = pkg.NativeHRNet(c, 17, (h, w), "fp32", max_batch=n, device=dev.index).load_state_dict(pkg.synth_state_dict(c, 17, 0))
    g = torch.Generator(device=dev).manual_seed(1234)
    images = torch.randn((n, 3, h, w), generator=g, device=dev, dtype=torch.float32)
    boxes = torch.from_numpy(pkg.synth_boxes(n, seed=100)).to(dev)
    net.predict_crops(images, boxes)
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        net.predict_crops(images, boxes)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    value = steps * n / el
    flops = net.flops_per_crop()
    conv_ms, other = net.profile_pass(images)
    conv_ms, other = net.profile_pass(images)
    infos = net.conv_infos()
    conv_tf = sum(i.flops for i in infos) * n / (sum(conv_ms) * 1e-3) / 1e12
    net.close()
    return {"workload": "HRNet-W48 384x288, batch=256 random crops, fp32 (v_mfma_f32_16x16x4_f32), model forward + decode: the headline "
                        "shape in the parity mode (joint coordinates identical to the CPU reference, parity.fp32_coords_identical)",
            "value": round(value, 1), "unit": "crops/s", "ms_per_step": round(el / steps * 1e3, 3), "dtype": "f32", "steps": steps,
            "whole_net_tflops": round(value * flops / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": "all convolutions of the pass (HIP events around every kernel of one pass)",
                         "achieved": round(conv_tf, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(conv_tf / PEAK_F32_TFLOPS, 4),
                         "whole_net_frac": round(value * flops / 1e12 / PEAK_F32_TFLOPS, 4)}}


def parity_measure(pkg, net, a, images, boxes, dev, cpu_out):
    """The first PARITY_CROPS crops of the timed batch, taken from a full-batch call (the batch-256 code path itself),
    against the CPU oracle's outputs for the same crops (computed by the cpu_baseline worker)."""
    import numpy as np
    import torch

    k = min(PARITY_CROPS, a.batch)
    ref = np.load(cpu_out)
    ref_hm, ref_pts = ref["hm"][:k], ref["pts"][:k]
    emu_hm = ref["emu_hm"][:k] if "emu_hm" in ref.files else None
    hm, pts = net.predict_crops(images, boxes, return_heatmaps=True)      # whole batch, as timed
    hm, pts = hm[:k].cpu().numpy(), pts[:k].cpu().numpy()
    del ref
    hw = ref_hm.shape[-1]
    flat, rflat = hm.reshape(k, hm.shape[1], -1), ref_hm.reshape(k, hm.shape[1], -1)
    am, ram = flat.argmax(-1), rflat.argmax(-1)
    cells = np.maximum(np.abs(am // hw - ram // hw), np.abs(am % hw - ram % hw))
    px = cells * 4                                                        # heat-map cell = 4 px of the crop
    edges = [0, 4, 8, 16, 32]
    hist = {"0": int((px == 0).sum())}
    for lo, hi in zip(edges, edges[1:]):
        hist["%d-%d" % (lo + 1, hi)] = int(((px > lo) & (px <= hi)).sum())
    hist[">32"] = int((px > 32).sum())
    out = {"subset": "first %d crops of the timed batch, results taken from the full batch-%d call (micro-batch %d)" % (k, a.batch, a.max_batch),
           "reference": "CPU oracle (oracle/hrnet_torch_oracle.py = the reference's device='cpu' path), fp32",
           "%s_argmax_agree_frac" % a.dtype: round(float((am == ram).mean()), 4),
           "%s_px_hist" % a.dtype: hist, "%s_px_unit" % a.dtype: "crop pixels, max(|dy|,|dx|) per (crop, joint), %d joints" % am.size,
           "%s_max_abs_dH" % a.dtype: round(float(np.abs(hm - ref_hm).max()), 5), "heatmap_sigma": round(float(ref_hm.std()), 4),
           "%s_coords_identical" % a.dtype: bool(np.array_equal(pts[..., :2], ref_pts[..., :2]))}
    if emu_hm is not None:
        # Is the bf16 engine's distance from the fp32 reference bf16 arithmetic, or the kernels?  The emulation IS bf16-storage
        # arithmetic (oracle/hrnet_torch_oracle.py: EngineEmulation, pinned to the reference tap by tap with its roundings off):
        # the engine against it, and the emulation's own agreement with the fp32 reference.
        eflat = emu_hm.reshape(k, hm.shape[1], -1)
        eam = eflat.argmax(-1)
        out["bf16_vs_emulation_max_rel"] = round(float(np.abs(hm - emu_hm).max() / np.abs(emu_hm).max()), 6)
        out["bf16_vs_emulation_max_abs_dH"] = round(float(np.abs(hm - emu_hm).max()), 5)
        out["bf16_vs_emulation_argmax_agree_frac"] = round(float((am == eam).mean()), 4)
        out["emulation_vs_fp32_argmax_agree_frac"] = round(float((eam == ram).mean()), 4)
        out["emulation_vs_fp32_max_abs_dH"] = round(float(np.abs(emu_hm - ref_hm).max()), 5)
        out["emulation"] = ("bf16-storage arithmetic restated on the CPU: folded weights and every stored activation rounded to "
                            "bf16, fp32 accumulation; per-stage pins: tests/test_bf16_pin.py")
    if a.dtype != "fp32" and a.model_name == "HRNet":   # the parity mode on the same crops
        f32 = pkg.NativeHRNet(a.c, 17, (a.height, a.width), "fp32", max_batch=k, device=dev.index).load_state_dict(pkg.synth_state_dict(a.c, 17, 0))
        hm32, pts32 = f32.predict_crops(images[:k], boxes[:k], return_heatmaps=True)
        hm32, pts32 = hm32.cpu().numpy(), pts32.cpu().numpy()
        f32.close()
        out["fp32_coords_identical"] = bool(np.array_equal(pts32[..., :2], ref_pts[..., :2]))
        out["fp32_max_abs_dH"] = float(np.abs(hm32 - ref_hm).max())
        out["fp32_max_coord_dev_px"] = float(np.abs(pts32[..., :2] - ref_pts[..., :2]).max())
    return out


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU over RCCL) and relay rank 0's line."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    # (HRN_BENCH_DEVICES="0,0": ranks share the listed devices -- the 2-rank test on a 1-GPU box; then the backend must be gloo)
    if have < a.gpus and not os.environ.get("HRN_BENCH_DEVICES"):
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (a.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / device-memory sharing across processes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stdout.write(r.stdout)
        raise SystemExit("the %d-rank run failed (exit code %d)" % (a.gpus, r.returncode))
    print(lines[-1], flush=True)


def main():
    a = parse()
    if a.cpu_worker:
        return _cpu_worker(a.cpu_worker)
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not launched:
        return self_launch(a)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    # rank -> device: LOCAL_RANK, or the HRN_BENCH_DEVICES list (test hook: two ranks on ONE GPU, which RCCL refuses -- the
    # backend is then HRN_BENCH_BACKEND=gloo and device tensors are staged through the host, as dist.ShardedHRNet does)
    devmap = [int(x) for x in os.environ["HRN_BENCH_DEVICES"].split(",")] if os.environ.get("HRN_BENCH_DEVICES") else None
    local = devmap[rank % len(devmap)] if devmap else local
    backend = os.environ.get("HRN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def reduce_max(x):       # a python float, MAX over the ranks
        if not dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(x):    # one python float per rank
        if not dist:
            return [x]
        mine = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = torch.empty((world,), dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(allr, mine)
        return [float(v) for v in allr.cpu()]

    pkg = importlib.import_module("simple-hrnet_amd")
    shard = importlib.import_module("simple-hrnet_amd.dist")

    net = pkg.NativeHRNet(a.c, 17, (a.height, a.width), a.dtype, max_batch=a.max_batch, device=local, model_name=a.model_name)
    eng = shard.ShardedHRNet(net, dist)
    sd0 = pkg.synth_state_dict(a.c, 17, 0, model=a.model_name) if rank == 0 else None
    if a.zeros and sd0 is not None:   # zero operands everywhere: weights, BN shift / mean (var 1 keeps the fold finite), biases
        sd0 = type(sd0)((k, (np.ones_like(v) if k.endswith("running_var") else np.zeros_like(v))) for k, v in sd0.items())
    eng.load_and_broadcast(sd0, src=0)
    del sd0

    if a.clip:   # only the configs[4] measurement
        cpu = None
        res, (clip, dets, mine_pts) = clip_measure(pkg, net, dist, rank, world, cpu)
        if a.check_gather and dist:
            # every rank holds the joints of ITS frames (zeros elsewhere): their sum over the ranks is the whole clip, which must equal
            # what ONE engine computes when it is dealt every frame (frames are independent; the sharding must not change a joint)
            t = torch.from_numpy(mine_pts).to(dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            alone, _ = run_clip(net, torch.from_numpy(clip).pin_memory(), dets, "per_frame", 0, 1)
            same = bool(np.array_equal(t.cpu().numpy(), alone))
            res["gathered_joints_equal_single_engine"] = reduce_max(0.0 if same else 1.0) == 0.0
            res["collective_backend"] = dist.get_backend()
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"clip": res}), flush=True)
        return

    # synthetic crops, device resident (post-normalisation domain ~N(0,1)), different per rank
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((a.batch, 3, a.height, a.width), generator=g, device=dev, dtype=torch.float32)
    if a.zeros:
        images.zero_()
    boxes_np = pkg.synth_boxes(a.batch, seed=100 + rank)
    boxes = torch.from_numpy(boxes_np).to(dev)

    # the timed path: ONE engine per GPU, ONE stream.  (Two engines sharing the GPU with half of the batch each measure +2.3 ... +3.3 %
    # in same-box pairs -- the `two_lanes` side measurement below -- and are NOT the headline: launches that overlap on the GPU cannot be
    # timed one by one, so `roofline` and the rocprofv3 kernel statistics would no longer describe the timed path.  VERDICT r4 item 5.)
    def step():
        return eng.predict_crops_local_then_gather(images, boxes)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pts = step()
    torch.cuda.synchronize()
    el_mine = time.perf_counter() - t0     # this rank's own loop (before the closing barrier)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    el = reduce_max(el)
    per_rank = gather_floats(a.batch * a.steps / el_mine)
    assert tuple(pts.shape) == (a.batch * world, 17, 3) and bool(torch.isfinite(pts).all())
    # --check-gather: the gathered joints of ALL ranks against ONE engine run over every rank's crops on this rank's GPU (crops and
    # boxes are seeded per rank, so any rank can regenerate them): what the sharding + all-gather must not change
    gather_ok = None
    if a.check_gather and world > 1:
        ref = []
        for r in range(world):
            gr = torch.Generator(device=dev).manual_seed(1234 + r)
            im = torch.randn((a.batch, 3, a.height, a.width), generator=gr, device=dev, dtype=torch.float32)
            ref.append(net.predict_crops(im, torch.from_numpy(pkg.synth_boxes(a.batch, seed=100 + r)).to(dev)))
        gather_ok = bool(torch.equal(torch.cat(ref, 0), pts))
        gather_ok = reduce_max(0.0 if gather_ok else 1.0) == 0.0

    out = None
    if rank == 0:
        crops_total = a.batch * world * a.steps
        value = crops_total / el
        flops = net.flops_per_crop()
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        name = "HRNet-W%d" % a.c if a.model_name == "HRNet" else "PoseResNet-%d" % a.c
        out = {
            "metric": "person-crops/sec %s %dx%d (model forward + heat-map decode)" % (name, a.height, a.width),
            "value": round(value, 2), "unit": "crops/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic" if not a.zeros else "ZEROS (diagnostic run: not the metric)",
            "config": {"workload": "%s %dx%d, batch=%d random crops per GPU, %s%s"
                                   % (name, a.height, a.width, a.batch, a.dtype,
                                      " MFMA (BASELINE configs[2])" if (a.model_name, a.c, a.dtype, world) == ("HRNet", 48, "bf16", 1)
                                      else " MFMA, sharded over %d GPUs (BASELINE configs[3] shape: %d crops per step)" % (world, a.batch * world)
                                      if (a.model_name, a.c, a.dtype) == ("HRNet", 48, "bf16") else ""),
                       "global_batch": a.batch * world, "micro_batch": a.max_batch,
                       "lanes_per_gpu": 1,
                       "parallelism": "dp%d (crop sharding, RCCL all-gather of keypoints)" % world,
                       "weights": "random-init seeded (synth_state_dict seed 0), BN stats randomised",
                       "engine_switches": net.switches()},
            "rccl_ranks": (dist.get_world_size() if dist.get_backend() == "nccl" else 0) if dist else 0,
            "collective_backend": (dist.get_backend() if dist else None),
            "gathered_joints_equal_single_engine": gather_ok,
            "per_rank_crops_per_s": [round(x, 1) for x in per_rank],
            "whole_net_tflops": round(value / world * flops / 1e12, 2),
            "gflop_per_crop": round(flops / 1e9, 3),
        }
        if not a.no_roofline and a.model_name == "HRNet":
            # per-kernel HIP-event times of one internal pass, same stream as the launches
            nb = min(a.max_batch, a.batch)
            for _ in range(2):
                conv_ms, other = net.profile_pass(images[:nb])
            infos = net.conv_infos()
            # dominant kernel = conv3x3_lds_kernel; graded subset = the stage-3/4 BasicBlock convs.  The k-th conv of
            # every branch of a stage module is ONE grouped launch: count launches by (module, block, conv) key.
            sub = [(i, ms) for i, ms in zip(infos, conv_ms)
                   if b".branches." in i.name and (i.name.startswith(b"stage3") or i.name.startswith(b"stage4"))]

            def _group(name):
                f = name.decode().split(".")     # stageX.M.branches.B.K.convN
                return (f[0], f[1], f[4], f[5])

            launches = len({_group(i.name) for i, _ in sub})
            sub_flops = sum(i.flops for i, _ in sub) * nb
            sub_ms = sum(ms for _, ms in sub)
            ach = sub_flops / (sub_ms * 1e-3) / 1e12
            # algorithmic HBM bytes of the same launches: input + output (+ residual) activations and the weights
            esz = 2 if a.dtype == "bf16" else 4
            sub_bytes = sum(((2 + i.has_residual) * i.cout * i.out_h * i.out_w * nb + 9 * i.cin * i.cout) * esz for i, _ in sub)
            traffic, traffic_src = pmc_traffic(a, nb)
            all_ms = sum(conv_ms) + sum(other.values())
            out["roofline"] = {
                "bound": "mfma",
                "kernel": "conv3x3_lds_kernel: stage-3/4 BasicBlock 3x3 convs, %d convs in %d grouped launches"
                          % (len(sub), launches),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_unit": "bytes per grouped launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(sub_bytes / launches),
                "flops_per_launch": round(sub_flops / launches / 1e9, 3), "flops_unit": "GFLOP (algorithmic, 2*MAC)",
                "avg_launch_ms": round(sub_ms / launches, 4),
                "timing": "HIP events on the launch stream around every kernel of one pass of %d crops" % nb,
                "subset_share_of_pass_time": round(sub_ms / all_ms, 3),
                "pass_ms": {"convs": round(sum(conv_ms), 3), **{k: round(v, 3) for k, v in other.items()}},
                "source_hash": source_hash(),
            }
        # Side measurement, never `value`: the same step with TWO engines on this GPU, each on its own stream with half of the
        # batch (native.MultiDeviceHRNet).  The launches of one engine fill the tails and
        # drains of the other's: same kernels, same joints, twice the activation workspace.
        if world == 1 and not a.no_two_lanes and a.model_name == "HRNet" and a.batch >= 4:
            try:
                native = importlib.import_module("simple-hrnet_amd.native")
                two = native.MultiDeviceHRNet([local, local], a.c, 17, (a.height, a.width), a.dtype,
                                              max_batch=min(a.max_batch, -(-a.batch // 2)), model_name=a.model_name).adopt_from(net)
                for _ in range(2):
                    p2 = two.predict_crops(images, boxes)
                torch.cuda.synchronize()
                n2 = max(3, min(a.steps, 10))
                t2 = time.perf_counter()
                for _ in range(n2):
                    p2 = two.predict_crops(images, boxes)
                torch.cuda.synchronize()
                v2 = a.batch * n2 / (time.perf_counter() - t2)
                out["two_lanes"] = {"crops_per_s": round(v2, 1), "vs_value": round(v2 / value, 4), "steps": n2,
                                    "same_joints_as_one_engine": bool(torch.equal(p2, pts)),
                                    "workspace_bytes_total": int(sum(e.workspace_bytes() for e in two.nets)),
                                    "note": "two engines on one GPU, %d crops each per step, on two streams" % (-(-a.batch // 2))}
                two.close()
            except Exception as e:   # a side measurement: never lose the main one to it
                out["two_lanes"] = {"error": "%s: %s" % (type(e).__name__, e)}
        cpu = None
        tmpdir = tempfile.mkdtemp(prefix="hrn_bench_")
        if world == 1 and not a.no_cpu_baseline and a.model_name == "HRNet":
            k = min(PARITY_CROPS, a.batch)
            spec = {"c": a.c, "h": a.height, "w": a.width, "budget_s": a.cpu_seconds, "crops": os.path.join(tmpdir, "crops.npy"),
                    "boxes": boxes_np[:k].tolist(), "out": os.path.join(tmpdir, "ref.npz"), "clip": None,
                    "emulate_bf16": a.dtype == "bf16" and a.model_name == "HRNet",
                    "peaked": None if a.no_peaked else {"n": a.batch, "seed": PEAKED_SEED, "idx": [i for i in PEAKED_IDX if i < a.batch]}}
            np.save(spec["crops"], images[:k].cpu().numpy())
            if not a.no_clip:
                clip, dets = make_clip()
                np.save(os.path.join(tmpdir, "frame0.npy"), clip[0])
                spec["clip"] = {"frame": os.path.join(tmpdir, "frame0.npy"), "dets": dets[0, :, :4].tolist(), "out": os.path.join(tmpdir, "clip_ref.npz")}
                del clip
            cpu = _run_cpu_worker(spec, a.cpu_seconds)
            if os.path.exists(spec["out"]):
                try:
                    out["parity"] = parity_measure(pkg, net, a, images, boxes, dev, spec["out"])
                except Exception as e:   # never lose the GPU number to the checker
                    out["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
                if not a.no_peaked:
                    try:
                        out["parity"]["peaked"] = peaked_measure(pkg, a, dev, spec["out"])
                    except Exception as e:
                        out["parity"]["peaked"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_clip and a.model_name == "HRNet":
            try:
                res, (clip, dets, gpu_pts) = clip_measure(pkg, net, None, 0, 1, (cpu or {}).get("clip"))
                ref_path = os.path.join(tmpdir, "clip_ref.npz")
                if os.path.exists(ref_path):   # frame 0 against the reference loop restated on the CPU (bf16 engine: agreement rate)
                    ref = np.load(ref_path)
                    h4 = net.resolution[0] // 4
                    res["frame0_boxes_identical_to_cpu_loop"] = bool(np.array_equal(net.preprocess_frame(clip[0], dets[0])[1], ref["boxes"]))
                    res["frame0_joints_agree_frac"] = round(float((np.abs(gpu_pts[0][..., :2] - ref["pts"][..., :2]).max(-1) <
                                                                   (ref["boxes"][:, 3] - ref["boxes"][:, 1]).reshape(-1, 1) / h4 * 0.5).mean()), 4)
                out["clip"] = res
            except Exception as e:
                out["clip"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_config1 and a.model_name == "HRNet":
            try:
                out["config1_fp32"] = config1_measure(pkg, dev)
            except Exception as e:
                out["config1_fp32"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_fp32_w48 and a.model_name == "HRNet":
            try:
                out["fp32_w48_384x288"] = fp32_w48_measure(pkg, a, dev)
            except Exception as e:
                out["fp32_w48_384x288"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not a.no_prepath:
            out["prepath"] = prepath_measure(pkg, net, dev)
            out["pcie_inclusive"] = pcie_measure(pkg, net, min(a.batch, a.max_batch), dev)
        if cpu is not None:
            cpu.pop("clip", None)
            out["cpu_baseline"] = cpu
        import shutil

        shutil.rmtree(tmpdir, ignore_errors=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
