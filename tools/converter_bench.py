#!/usr/bin/env python3
"""The converter end to end: the synthetic 3 M x 2504 cohort as 22 BGZF files `chr{N}.filtered.vcf.gz` in /dev/shm ->
`VCFtoHDF5Converter(...).run()` -> OUT/{cohort}.h5 (the reference's CLI path, vcf_to_h5.py:182-232): seconds for the
whole run, for the engine + working store, and for the export of the .h5.  usage: python tools/converter_bench.py
[--variants N] [--chroms 1,2,...] [--outdir DIR] [--keep]"""
import argparse
import json
import logging
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=3_000_000)
    ap.add_argument("--chroms", default=",".join(str(c) for c in range(1, 23)))
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--outdir", default="")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    from haplohyped_varawareml_amd import store as store_mod
    from haplohyped_varawareml_amd.reader import write_bgzf_native
    from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
    logging.basicConfig(level=logging.WARNING)
    base = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    vcf_dir = os.path.join(base, "vcf")
    out_dir = a.outdir or os.path.join(base, "out")
    os.makedirs(vcf_dir)
    os.makedirs(out_dir, exist_ok=True)
    chroms = [int(c) for c in a.chroms.split(",")]
    sizes = synth.shard_sizes(a.variants)
    ctx = dev.Context(0)
    t0 = time.time()
    V = text_bytes = 0
    names = None
    for c in chroms:
        tab = synth.variant_table(1000 + c, sizes[c - 1], a.samples)
        t, n = ctx.synth_fixed(f"chr{c}", tab, a.samples, seed=1000 + c)
        host = torch.empty(n, dtype=torch.uint8).pin_memory()
        host.copy_(t)
        del t
        write_bgzf_native(os.path.join(vcf_dir, f"chr{c}.filtered.vcf.gz"), host.numpy(), level=6)
        if names is None:
            head = bytes(host[:1 << 20].numpy())
            line = [l for l in head.split(b"\n") if l.startswith(b"#CHROM")][0]
            names = [x.decode() for x in line.split(b"\t")[9:]]
        V += sizes[c - 1]
        text_bytes += n
    ctx.close()
    torch.cuda.empty_cache()
    prep = time.time() - t0
    sl = os.path.join(base, "samples.txt")
    open(sl, "w").write("\n".join(names) + "\n")
    file_bytes = sum(os.path.getsize(os.path.join(vcf_dir, f)) for f in os.listdir(vcf_dir))
    # time the export separately
    t_export = [0.0]
    real_export = store_mod.export_h5

    def timed_export(*args, **kw):
        t = time.perf_counter()
        r = real_export(*args, **kw)
        t_export[0] = time.perf_counter() - t
        return r
    store_mod.export_h5 = timed_export
    conv = VCFtoHDF5Converter("cohort", vcf_dir, out_dir, sl, cores=0, cxx_threads=1, n_gpus=1)
    t = time.perf_counter()
    h5 = conv.run()
    total = time.perf_counter() - t
    out = dict(variants=V, samples=a.samples, files=len(chroms), text_bytes=text_bytes, file_bytes=file_bytes,
               h5_bytes=os.path.getsize(h5), seconds=round(total, 3), seconds_engine_and_store=round(total - t_export[0], 3),
               seconds_export_h5=round(t_export[0], 3), variants_per_s=V / total,
               variants_per_s_engine_and_store=V / max(total - t_export[0], 1e-9), out_dir=out_dir, prep_seconds=round(prep, 1),
               ratio=V * a.samples * 2 / os.path.getsize(h5))
    print(json.dumps(out))
    if not a.keep:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
