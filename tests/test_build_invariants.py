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
    assert len(chain) == 4   # plain, projection shortcut (DS), 3x3 in front (C3 = 1), 3x3 in front of the last block (C3 = 2)
    for name, use in chain.items():
        assert use["ScratchSize"] == 0 and use.get("VGPRs Spill", 0) == 0, (name, use)


# ------------------------------------------------------------------------------------------------------------------
# Disassembly guards for the hand-scheduled assumptions of the 96-cout form (conv3x3_n96.inc) and the stride-2 slab kernel
# (conv_s2.hip): a silent miscompile on the next hipcc becomes a red test.
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _disassemble(source, tmp_path):
    """-> {kernel symbol: [(address, mnemonic + operands)]} of the gfx950 device code of `source`"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists(OBJDUMP):
        pytest.skip("no hipcc / llvm-objdump")
    obj = os.path.join(tmp_path, "dev.o")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "--no-gpu-bundle-output",
                          "-c", os.path.join(CSRC, source), "-o", obj], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = subprocess.run([OBJDUMP, "-d", obj], capture_output=True, text=True, timeout=300).stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None and "//" in line and line[:1] in " \t":
            ins, _, tail = line.partition("//")
            m = re.match(r"\s*([0-9A-F]+):", tail)
            if m and ins.strip():
                cur.append((int(m.group(1), 16), re.sub(r"\s+", " ", ins).strip()))
    return kernels


@pytest.fixture(scope="module")
def c3_asm(tmp_path_factory):
    return _disassemble("conv3x3_lds.hip", str(tmp_path_factory.mktemp("c3")))


def _main_kernel(c3_asm):
    names = [k for k in c3_asm if "conv3x3_lds_kernelILi48ELi3" in k]
    assert len(names) == 1
    return c3_asm[names[0]]


def test_n96_counted_wait_table_is_24_entries_of_8_bytes(c3_asm):
    """n96_vmcnt_dyn jumps to  A + 20 + 8 * n  (A = the address s_getpc_b64 returns): every use must be followed by exactly
    five 4-byte scalar instructions and 24 (s_waitcnt vmcnt(k); s_branch END) pairs of 8 bytes with one common END."""
    ins = _main_kernel(c3_asm)
    sites = [i for i, (_, t) in enumerate(ins) if t.startswith("s_getpc_b64 s[100:101]")]
    assert len(sites) >= 2
    for i in sites:
        a = ins[i][0] + 4
        head = [t.split()[0] for _, t in ins[i + 1:i + 6]]
        assert head == ["s_lshl_b32", "s_add_u32", "s_add_u32", "s_addc_u32", "s_setpc_b64"], head
        assert ins[i + 1][1].endswith(", 3") and ins[i + 2][1].endswith(", 20")          # 8-byte entries, table at A + 20
        assert [ad for ad, _ in ins[i + 1:i + 6]] == [a + 4 * k for k in range(5)]
        ends = set()
        for k in range(24):
            (aw, tw), (ab, tb) = ins[i + 6 + 2 * k], ins[i + 7 + 2 * k]
            assert aw == a + 20 + 8 * k and tw == "s_waitcnt vmcnt(%d)" % k, (k, hex(aw), tw)
            assert ab == aw + 4 and tb.startswith("s_branch "), (k, tb)
            ends.add(ab + 4 + 4 * int(tb.split()[1]))
        assert ends == {a + 20 + 8 * 24}, ends                                            # every entry leaves to just behind the table


def test_n96_s100_s101_live_only_inside_the_counted_wait(c3_asm):
    """the computed jump builds its target in s[100:101] (declared clobbered): no other instruction may touch them"""
    ins = _main_kernel(c3_asm)
    allowed = ("s_getpc_b64 s[100:101]", "s_add_u32 s100, s100,", "s_addc_u32 s101, s101, 0", "s_setpc_b64 s[100:101]")
    for _, t in ins:
        if re.search(r"\bs10[01]\b|s\[100:101\]|s\[100:10[2-9]\]|s\[9[6-9]:10[0-9]\]", t):
            assert t.startswith(allowed), t


def test_only_lds_dma_reads_m0_in_the_basicblock_kernel(c3_asm):
    """n96_glds writes M0 from inline asm without saving it: correct as long as the only M0 readers of the kernel are LDS-DMA
    instructions, each of which has its own M0 write shortly in front (the asm's s_mov, or the compiler's for its builtins)."""
    ins = _main_kernel(c3_asm)
    m0_readers = ("global_load_lds", "buffer_load", "ds_gws", "s_sendmsg", "s_movrel", "v_movrel", "v_interp", "ds_add_gs", "ds_sub_gs",
                  "ds_read_addtid", "ds_write_addtid", "s_ttrace")
    n_dma = 0
    for i, (_, t) in enumerate(ins):
        op = t.split()[0]
        operands = t.replace(",", " ").split()[1:]
        if "m0" in operands:
            assert op.startswith("s_") and operands[0] == "m0" and "m0" not in operands[1:], t   # M0 is only ever WRITTEN (scalar ALU)
        if op.startswith(m0_readers):
            assert op.startswith("global_load_lds"), t
            n_dma += 1
            prev = [x for _, x in ins[max(0, i - 48):i]]   # (the asm's own s_mov sits 2 back; the compiler's may be hoisted a little)
            assert any(re.match(r"s_\w+ m0,", x) for x in prev), (t, prev[-4:])
    assert n_dma > 20


def test_s2_slab_kernel_register_and_store_invariants(tmp_path):
    """conv_s2.hip keeps up to 168 weight registers live and leaves its last stores in flight across the tile barrier with a
    COUNTED vmcnt (stores per fragment x fragments): no spills, two waves per SIMD, and exactly those store instructions."""
    kernels = _resource_usage("conv_s2.hip", str(tmp_path))
    (name, use), = [(k, v) for k, v in kernels.items() if "conv_s2_slab_kernel" in k]
    assert use["ScratchSize"] == 0 and use.get("VGPRs Spill", 0) == 0 and use["VGPRs"] <= 256 and use["Occupancy"] >= 2, use
    asm = _disassemble("conv_s2.hip", str(tmp_path))
    (ins,) = [v for k, v in asm.items() if "conv_s2_slab_kernel" in k]
    ops = [t.split()[0] for _, t in ins]
    # three instantiations in the kernel -- <48, 3, 2>: per fragment one 16-byte + one 8-byte store; <32, 2, 2> and <64, 2, 2>: one
    # 16-byte store per fragment; two fragments each: these are the counts the counted waits (vmcnt 4 / 2 / 1) stand on
    assert ops.count("global_store_dwordx4") == 6 and ops.count("global_store_dwordx2") == 2, [o for o in ops if "store" in o]
    # one fully unrolled K loop per instantiation, nothing duplicated: 14 x 3 x 2 + 9 x 2 x 2 + 18 x 2 x 2
    assert sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 84 + 36 + 72
    # the stores come after every LDS-DMA of the tile loop's flush: the last LDS-DMA before the first store is followed by no other
    first_store = ops.index("global_store_dwordx4")
    assert any(o.startswith("global_load_lds") for o in ops[:first_store])
