"""Static instruction counts per marked section of k_lz4_bitplanes<D> (hipcc -S -DBP_MARKS): which part of the
window loop the vector instructions sit in.  usage: python tools/isa_sections.py [depth]   (needs hipcc)
Counts are per textual section between markers, in listing order (a basic block the compiler moved elsewhere is
attributed to where it landed), so read them as a map, not as a cycle count."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # 1: the plane-input instantiation, 0: int8 input
out = "/tmp/lz4bits_marks.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-DBP_MARKS",
                       "-I", os.path.join(ROOT, "include"), "-o", out,
                       os.path.join(ROOT, "haplohyped_varawareml_amd", "csrc", "lz4bits.hip")], stderr=subprocess.DEVNULL)
inside, sec = False, "entry"
cnt = collections.OrderedDict()
for line in open(out):
    if line.startswith(f"_Z15k_lz4_bitplanesILi{depth}ELb{planes}E"):
        inside = True
        continue
    if not inside:
        continue
    t = line.strip()
    if t.startswith("s_endpgm") and sec == "loop_done":
        pass
    if t.startswith(".section") or t.startswith(".rodata"):
        break
    m = re.match(r"; ==MARK (\w+)", t)
    if m:
        sec = m.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
        "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other"
    cnt.setdefault(sec, collections.Counter())[kind] += 1
    if op == "s_waitcnt" and "lgkmcnt" in t:
        cnt[sec]["lgkm_wait"] += 1          # a wait on LDS / scalar-memory results: one link of the wave's dependent chain
print(f"k_lz4_bitplanes<{depth}>: instructions in the listing after each marker")
for k, c in cnt.items():
    print(f"  {k:12s} valu {c['valu']:5d}  salu {c['salu']:5d}  lds {c['lds']:4d}  vmem {c['vmem']:4d}  waits on lds {c['lgkm_wait']:3d}")
