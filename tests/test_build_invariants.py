"""Compile-time invariants of the hand-scheduled kernels.  The BasicBlock kernel issues loads from inline asm and waits
for them with counted ``s_waitcnt``s: the compiler does not know those registers are still in flight, so a register
allocator that SPILLS one of them right after the load would store garbage -- silently.  The build therefore has to
stay spill-free for these kernels; hipcc reports it (``-Rpass-analysis=kernel-resource-usage``)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simple-hrnet_amd", "csrc")


def _resource_usage(source, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-c",
                          os.path.join(CSRC, source), "-o", os.path.join(tmp_path, "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = int(m.group(2))
    return kernels


def test_basicblock_kernels_do_not_spill(tmp_path):
    kernels = _resource_usage("conv3x3_lds.hip", str(tmp_path))
    lds = {k: v for k, v in kernels.items() if "conv3x3_lds_kernel" in k}
    assert len(lds) == 4                                           # <48,3> <32,4> <32,3> <32,2>
    for name, use in lds.items():
        assert use["ScratchSize"] == 0 and use.get("VGPRs Spill", 0) == 0, (name, use)
        assert use["VGPRs"] <= 256 and use["Occupancy"] >= 2, (name, use)   # two waves per SIMD: 8 waves share a CU's LDS


def test_chain_kernel_does_not_spill(tmp_path):
    kernels = _resource_usage("bottleneck_chain.hip", str(tmp_path))
    chain = {k: v for k, v in kernels.items() if "bottleneck_chain_kernel" in k}
    assert len(chain) == 2
    for name, use in chain.items():
        assert use["ScratchSize"] == 0 and use.get("VGPRs Spill", 0) == 0, (name, use)
