#!/usr/bin/env python3
"""End-to-end (inflate- and PCIe-inclusive) rate of the native ingest engine (csrc/ingest.hip): synthetic
per-chromosome shards of the 3 M x 2504 cohort as files in /dev/shm (BGZF at --level, plain gzip or uncompressed)
or as text in pinned host memory -> framed chunks back on the host.  Reported next to the HBM-resident number of
bench.py (which runs the same legs on a bounded sample by default); never bench.py's `value`."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_shards(ctx, chroms, variants, samples, kind, level, tmpdir):
    """-> [(path or pinned tensor, region, n_variants, text_bytes)]"""
    import torch
    from haplohyped_varawareml_amd import synth
    from haplohyped_varawareml_amd.reader import write_bgzf_native
    sizes = synth.shard_sizes(variants)
    out = []
    for c in chroms:
        V = sizes[c - 1]
        tab = synth.variant_table(1000 + c, V, samples)
        t, n = ctx.synth_fixed(f"chr{c}", tab, samples, seed=1000 + c)
        host = torch.empty(n, dtype=torch.uint8).pin_memory()
        host.copy_(t)
        del t
        if kind == "memory":
            out.append((host, f"chr{c}", V, n))
            continue
        p = os.path.join(tmpdir, f"chr{c}.filtered.vcf.gz")
        if kind == "bgzf":
            write_bgzf_native(p, host.numpy(), level=level)
        elif kind == "gzip":
            import gzip
            with gzip.open(p, "wb", compresslevel=level) as f:
                f.write(host.numpy().tobytes())
        else:
            host.numpy().tofile(p)
        out.append((p, f"chr{c}", V, n))
    torch.cuda.empty_cache()
    return out


def open_engine(ctx, device_inflate, threads, block_mb, files_ahead=1, fmt=None, expect_samples=0):
    from haplohyped_varawareml_amd import device as dev
    from haplohyped_varawareml_amd.ingest import Ingest
    t0 = time.perf_counter()
    ing = Ingest(ctx, fmt=fmt or dev.BLOSC2, device_inflate=device_inflate, n_threads=threads, block_bytes=block_mb << 20,
                 files_ahead=files_ahead, expect_samples=expect_samples)
    return ing, time.perf_counter() - t0


def run(ing, shards):
    """one pass of all shards through an open engine (its buffers are warm after the first pass: what a converter
    sees from its second file on)"""
    from haplohyped_varawareml_amd.ingest import Columns, InputEnd
    t0 = time.perf_counter()
    for src, region, _, _ in shards:
        ing.add_file(src, region) if isinstance(src, str) else ing.add_memory(src, region)
    framed = kept = 0
    per_file = []
    for ev in ing.events():
        if isinstance(ev, Columns):
            framed += ev.framed.size
        elif isinstance(ev, InputEnd):
            kept += ev.stats["n_kept"]
            per_file.append(round(ev.stats["seconds"], 4))
            if len(per_file) == len(shards):
                break
    t1 = time.perf_counter()
    return dict(seconds=t1 - t0, framed_bytes=framed, n_kept=kept, per_file_seconds=per_file)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=3_000_000, help="cohort size the shard sizes are taken from")
    ap.add_argument("--chroms", default="1,2,3,4", help="which shards (chromosome numbers) to run")
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--kind", choices=["bgzf", "gzip", "plain", "memory"], default="bgzf")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--block-mb", type=int, default=0, help="text block size (0: the engine default)")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--device-inflate", action="store_true", help="BGZF members inflated on the device (f-4)")
    ap.add_argument("--level", type=int, default=6, help="zlib level of the synthetic BGZF / gzip files")
    ap.add_argument("--files-ahead", type=int, default=1)
    ap.add_argument("--no-expect", action="store_true", help="do not tell the engine the sample count at open (round 3's behaviour)")
    a = ap.parse_args()
    if os.environ.get("HHGT_GC_DEBUG"):     # collector pauses of the consuming thread (the engine's out slots wait on it)
        import gc
        t_gc = [0.0]

        def on_gc(phase, info):
            if phase == "start":
                t_gc[0] = time.perf_counter()
            elif time.perf_counter() - t_gc[0] > 1e-3:
                print(f"[gc] generation {info['generation']}: {(time.perf_counter() - t_gc[0]) * 1e3:.1f} ms", file=sys.stderr)
        gc.callbacks.append(on_gc)
    import torch  # noqa: F401
    from haplohyped_varawareml_amd import device as dev
    ctx = dev.Context(0)
    chroms = [int(x) for x in a.chroms.split(",")]
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t0 = time.time()
    shards = make_shards(ctx, chroms, a.variants, a.samples, a.kind, a.level, d)
    prep = time.time() - t0
    ing, t_open = open_engine(ctx, a.device_inflate, a.threads, a.block_mb, a.files_ahead, expect_samples=0 if a.no_expect else a.samples)
    runs = [run(ing, shards) for _ in range(a.repeat)]
    ing.close()
    best = min(runs[1:] or runs, key=lambda r: r["seconds"])     # the first pass also pins the engine's staging memory
    V = sum(s[2] for s in shards)
    text = sum(s[3] for s in shards)
    fbytes = sum(os.path.getsize(s[0]) for s in shards if isinstance(s[0], str))
    out = dict(kind=a.kind, device_inflate=bool(a.device_inflate and a.kind == "bgzf"), level=a.level, chroms=chroms,
               variants=V, samples=a.samples, text_bytes=text, file_bytes=fbytes, seconds=best["seconds"],
               all_seconds=[round(r["seconds"], 4) for r in runs], first_pass_seconds=round(runs[0]["seconds"], 4),
               engine_open_seconds=round(t_open, 4),
               variants_per_s=V / best["seconds"], text_GBps=text / best["seconds"] / 1e9,
               ratio=V * a.samples * 2 / max(best["framed_bytes"], 1), per_file_seconds=best["per_file_seconds"],
               host_cores=os.cpu_count(), threads=a.threads, block_mb=a.block_mb, prep_seconds=round(prep, 2))
    assert best["n_kept"] == V, (best["n_kept"], V)
    print(json.dumps(out))
    for s in shards:
        if isinstance(s[0], str):
            os.remove(s[0])
    os.rmdir(d)
    ctx.close()


if __name__ == "__main__":
    main()
