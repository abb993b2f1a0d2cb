#!/usr/bin/env python3
"""Compress stage alone on one shard of the bench cohort (default chr1 of 3 M x 2504: 1.3 GB of G, 318 k planes):
kernel development driver (tools/pmc_kernel.sh profiles it).  Prints ms per launch, ratio and what the stage timers saw."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=3_000_000)
    ap.add_argument("--chrom", type=int, default=1)
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--clevel", type=int, default=5)
    ap.add_argument("--no-verify", action="store_true", help="development builds whose streams are not valid (timing only)")
    a = ap.parse_args()
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    ctx = dev.Context(0)
    ctx.set_clevel(a.clevel)
    V = synth.shard_sizes(a.variants)[a.chrom - 1]
    seed = 1000 + a.chrom
    tab = synth.variant_table(seed, V, a.samples)
    text, n = ctx.synth_fixed(f"chr{a.chrom}", tab, a.samples, seed=seed)
    lay = dev.make_layout(a.samples, V)
    res = ctx.encode_text(text, a.samples, region=f"chr{a.chrom}", layout=lay)
    ctx.pad_tail(res)
    chunk = lay.sc * lay.vc * 2
    dst, off, total = ctx.compress(res.G, chunk)
    if not a.no_verify:
        back, bad = ctx.decompress(dst, off, res.G.numel() // chunk, chunk)
        assert bad == 0 and torch.equal(back, res.G)
    ctx.profile(True)
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ctx.compress(res.G, chunk, dst=dst, chunk_off=off, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    st = ctx.profile_read()
    print(json.dumps(dict(variants=V, samples=a.samples, planes=res.G.numel() // 4096, raw_bytes=res.G.numel(), ms_per_launch=dt * 1e3,
                          GBps_in=res.G.numel() / dt / 1e9, ratio=V * a.samples * 2 / total, clevel=a.clevel,
                          stages_ms={k: v["ms"] / a.reps for k, v in st.items()})))


if __name__ == "__main__":
    main()
