#!/usr/bin/env python3
"""D2H copy rate into pinned memory: alone, beside a busy compute stream, beside an H2D stream (development probe:
what the ingest engine's shipper can expect)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haplohyped_varawareml_amd import device as dev

ctx = dev.Context(0)
n = 256 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s_copy, s_busy, s_h2d = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
src = torch.randint(0, 2, (64 * 8192 * 2 * 400,), dtype=torch.uint8, device="cuda")


def d2h(reps=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s_copy):
        for _ in range(reps):
            h.copy_(d, non_blocking=True)
    s_copy.synchronize()
    return reps * n / (time.perf_counter() - t0) / 1e9


out = {"d2h_alone_GBps": d2h()}
with torch.cuda.stream(s_busy):
    for _ in range(30):
        ctx.compress(src, 64 * 8192 * 2, sync=False)
out["d2h_beside_lz4_GBps"] = d2h()
torch.cuda.synchronize()
with torch.cuda.stream(s_h2d):
    for _ in range(16):
        d2.copy_(h2, non_blocking=True)
out["d2h_beside_h2d_GBps"] = d2h()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s_h2d):
    for _ in range(8):
        d2.copy_(h2, non_blocking=True)
s_h2d.synchronize()
out["h2d_alone_GBps"] = 8 * n / (time.perf_counter() - t0) / 1e9
print(json.dumps(out))
