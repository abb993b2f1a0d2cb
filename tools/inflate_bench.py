#!/usr/bin/env python3
"""f-4: BGZF members inflated on the device — kernel rate on a synthetic fixed-width VCF shard, next to zlib on one host core.
usage: python tools/inflate_bench.py [variants=30000] [level=6]"""
import json, os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haplohyped_varawareml_amd import device as dev, synth

V = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
S = 2504
ctx = dev.Context(0)
tab = synth.variant_table(1001, V, S)
text_dev, _ = ctx.synth_fixed("chr1", tab, S, seed=1001)
text = text_dev.cpu().numpy().tobytes()
t0 = time.perf_counter()
members = []
for i in range(0, len(text), 0xFF00):
    c = text[i:i + 0xFF00]
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(c) + co.flush()
    members.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
raw = b"".join(members)
t_deflate = time.perf_counter() - t0
t0 = time.perf_counter()
tabm = dev.bgzf_scan(raw)
t_scan = time.perf_counter() - t0
t0 = time.perf_counter()
n = 0
for o, l in zip(tabm["comp_off"][:2000], tabm["comp_len"][:2000]):
    n += len(zlib.decompress(raw[int(o):int(o) + int(l)], -15))
t_cpu = time.perf_counter() - t0
out, bad = ctx.inflate_bgzf(raw)
assert bad == 0 and torch.equal(out, text_dev[:out.numel()]) and out.numel() == len(text)
ctx.profile(True)
ctx.profile_reset()
N = 3
t0 = time.perf_counter()
for _ in range(N):
    out, bad = ctx.inflate_bgzf(raw)
torch.cuda.synchronize()
t_call = (time.perf_counter() - t0) / N
kern = ctx.profile_read()["inflate"]["ms"] / N   # inflate + CRC-32 kernels
print(json.dumps(dict(variants=V, samples=S, level=level, text_GB=len(text) / 1e9, bgzf_GB=len(raw) / 1e9, members=len(members),
                      ms_kernel=kern, text_GBps_kernel=len(text) / kern / 1e6, ms_call_with_upload=t_call * 1e3,
                      host_scan_ms=t_scan * 1e3, zlib_one_core_GBps=n / t_cpu / 1e9,
                      variants_per_s_kernel=V / (kern * 1e-3))))
