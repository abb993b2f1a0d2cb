#!/usr/bin/env python3
"""Stage times of the hot path on the non-headline configs (BASELINE configs 2 and 4): bench.py measures
config 3; this prints the same per-stage breakdown for the others (not the driver's metric)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["c2", "c4"], default="c4")
    ap.add_argument("--variants", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--keep-multiallelic", action="store_true",
                    help="the labelled NON-REFERENCE filter mode (multi-allelic SNP sites kept, allele indices > 1)")
    a = ap.parse_args()
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    ctx = dev.Context(0)
    ctx.set_keep_multiallelic(a.keep_multiallelic)
    if a.config == "c2":
        S, V, contig, seed = 1000, a.variants or 50_000, "chr22", 22
        tab = synth.variant_table(seed, V, S)
        text, n = ctx.synth_fixed(contig, tab, S, seed=seed)
    else:
        S, V, contig, seed = 5000, a.variants or 100_000, "chr4", 4
        tab = synth.mixed_table(seed, V, S)
        text, n, _ = ctx.synth_mixed(contig, tab, S, seed=seed)
    lay = dev.make_layout(S, V)
    res = ctx.encode_text(text, S, region=contig, layout=lay)
    chunk_nbytes = lay.sc * lay.vc * 2
    ctx.pad_tail(res)
    dst, off, total = ctx.compress(res.G, chunk_nbytes)
    ctx.profile(True)
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.encode_text(text, S, region=contig, v_base=0, out=res)
        ctx.pad_tail(res)
        ctx.compress(res.G, chunk_nbytes, dst=dst, chunk_off=off, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    st = ctx.profile_read()
    print(json.dumps(dict(config=a.config, mode="keep_multiallelic (non-reference)" if a.keep_multiallelic else "reference filter", variants=V, samples=S, text_bytes=n, kept=res.n_kept,
                          general_lines=res.stats["n_general_lines"], ms_per_step=dt * 1e3,
                          variants_per_s=V / dt, ratio=res.G.numel() / int(off[-1].item()),
                          stages_ms={k: v["ms"] / a.steps for k, v in st.items()})))


if __name__ == "__main__":
    main()
