#!/usr/bin/env python3
"""Read side: hhgt_decompress_chunks (LZ4 decode + un-shuffle) on one chr1-sized shard of the bench workload."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haplohyped_varawareml_amd import device as dev, synth

V, S = int(sys.argv[1]) if len(sys.argv) > 1 else 230_000, 2504
ctx = dev.Context(0)
tab = synth.variant_table(1001, V, S)
text, _ = ctx.synth_fixed("chr1", tab, S, seed=1001)
res = ctx.encode_text(text, S, region="chr1", layout=dev.make_layout(S, V))
ctx.pad_tail(res)
chunk = res.layout.sc * res.layout.vc * 2
n_chunks = res.G.numel() // chunk
dst, off, total = ctx.compress(res.G, chunk, typesize=2, blocksize=dev.DEFAULT_BLOCKSIZE, fmt=dev.BLOSC2)
back, bad = ctx.decompress(dst, off, n_chunks, chunk, typesize=2, blocksize=dev.DEFAULT_BLOCKSIZE)
assert bad == 0 and torch.equal(back, res.G)
torch.cuda.synchronize()
ctx.profile(True)
ctx.profile_reset()
t0 = time.perf_counter()
N = 5
for _ in range(N):
    back, bad = ctx.decompress(dst, off, n_chunks, chunk, typesize=2, blocksize=dev.DEFAULT_BLOCKSIZE)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
kern = ctx.profile_read().get("decode", {"ms": 0.0})["ms"] / N
print(json.dumps(dict(variants=V, samples=S, raw_GB=res.G.numel() / 1e9, compressed_GB=total / 1e9, ms_call=dt * 1e3, ms_kernel=kern,
                      out_GBps_kernel=res.G.numel() / (kern * 1e-3) / 1e9 if kern else None, variants_per_s=V / dt)))
