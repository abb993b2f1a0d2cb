"""CPU: code-generation properties of csrc/lz4.hip that the measured numbers depend on and that have broken before —
checked on the gfx950 assembly hipcc emits (cross-compiled, no GPU needed).

* the greedy-parse loop is inline asm with a read-write operand (the shrinking candidate mask m) next to an input
  holding the same value on entry (M): without an early-clobber mark the register allocator may give both the same
  SGPR pair, `s_and_b64 m, m, M` becomes `m & m` and the loop never ends (it did, once, on the GPU box);
* the per-wave sequence queue must be reached with LDS instructions (a pointer that loses its address space turns
  into flat_load / flat_store: same result, markedly slower);
* the default instantiation must fit the 7-waves-per-SIMD register budget with at most a handful of spills.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "haplohyped_varawareml_amd", "csrc", "lz4.hip")


@pytest.fixture(scope="module")
def lz4_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "lz4.s"
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.dirname(SRC), "-S", "--cuda-device-only", "-o", str(out), SRC],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def kernels(asm):
    """kernel name -> body text"""
    out = {}
    for m in re.finditer(r"^(_Z12k_lz4_blocksILi\d+ELi\d+ELb[01]EE\w+):\s*;.*?$", asm, re.M):
        end = asm.index("s_endpgm", m.end())
        out[m.group(1)] = asm[m.end():end]
    return out


def test_parse_loop_masks_do_not_alias(lz4_asm):
    ks = kernels(lz4_asm)
    assert len(ks) >= 6
    n_loops = 0
    for name, body in ks.items():
        for m in re.finditer(r"s_bitset1_b64 .*?\n(?:.*\n){1,12}?\s*s_and_b64 (s\[\d+:\d+\]), (s\[\d+:\d+\]), (s\[\d+:\d+\])", body):
            n_loops += 1
            dst, a, b = m.groups()
            assert dst == a and a != b, f"{name}: parse loop computes {m.group(0).splitlines()[-1].strip()} — m and M share registers"
    assert n_loops >= 5      # every window-encoder instantiation has the loop


def test_queue_stays_in_lds_and_registers_fit(lz4_asm):
    ks = kernels(lz4_asm)
    default = next(b for n, b in ks.items() if "ILi7ELi6ELb0E" in n)
    assert "flat_store" not in default and "flat_load" not in default, "sequence queue reached through flat addressing"
    assert "ds_write_b64" in default                      # the exec-masked enqueue
    meta = lz4_asm[lz4_asm.index("amdhsa.kernels"):]
    i = meta.index("k_lz4_blocksILi7ELi6ELb0E")
    chunk = meta[i:i + 1200]
    vgpr = int(re.search(r"\.vgpr_count:\s*(\d+)", chunk).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", chunk).group(1))
    assert vgpr <= 72 and spill <= 8, (vgpr, spill)
