#!/usr/bin/env python3
"""bench.py — variants/s, encode + compress, on synthetic 1000G-style VCF text resident in HBM.

One "step" = one pass of the hot path over the whole workload: for each of the 22 per-chromosome
shards (3 000 000 variants x 2504 samples, BASELINE.json configs[2] — the configuration the metric
is quoted on; ~30 GB of text, fits one MI355X):
    hhgt_encode_text  (line index -> fixed columns/filter -> GT tiles)  ->  int8 G, chunk-tiled
    hhgt_pad_tail
    hhgt_compress_chunks (byte-shuffle + LZ4 -> Blosc2-framed chunks)
Inputs (raw VCF text) are generated ON the GPU (csrc/synth.hip) before the timed region.

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks
time); shards are independent, there is no data-path collective.  Default "weak": every rank encodes
its own 3 M-variant cohort (different seeds).  --scaling strong splits the 22 shards of ONE cohort
over the ranks (longest-processing-time-first), as the north star's per-chromosome sharding does.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from haplohyped_varawareml_amd import device as dev  # noqa: E402
from haplohyped_varawareml_amd import synth  # noqa: E402
from haplohyped_varawareml_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variants", type=int, default=3_000_000)
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--vc", type=int, default=dev.DEFAULT_VC, help="variants per chunk (chunk = 64 x vc x 2 bytes)")
    ap.add_argument("--blocksize", type=int, default=dev.DEFAULT_BLOCKSIZE, help="Blosc2 block bytes")
    ap.add_argument("--clevel", type=int, default=5, help="codec level (reference: 5); 1-2 = run-only fast mode")
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no encode/compress overlap")
    ap.add_argument("--lookahead", type=int, default=1, help="shards the encode stream runs ahead of the compress stream")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo lets two ranks share one GPU in rehearsals")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline work")
    ap.add_argument("--cpu-extended", action="store_true", help="also time the all-cores and the reference-shaped CPU baselines")
    return ap.parse_args()


class Shard:
    pass


def build_shards(ctx, args, rank, world):
    sizes = synth.shard_sizes(args.variants)
    mine, seed_off = sharding.plan(sizes, rank, world, args.scaling)
    S = args.samples
    shards = []
    for ci in mine:
        V = sizes[ci]
        if V == 0:
            continue
        contig = f"chr{ci + 1}"
        seed = 1000 + (ci + 1) + seed_off
        tab = synth.variant_table(seed, V, S)
        text, nbytes = ctx.synth_fixed(contig, tab, S, seed=seed)
        sh = Shard()
        sh.contig, sh.V, sh.text, sh.nbytes = contig, V, text, nbytes
        sh.layout = dev.make_layout(S, V, vc=args.vc)
        cap = sh.layout.v_capacity
        d = ctx.device
        sh.res = dev.EncodeResult(torch.zeros(dev.layout_bytes(sh.layout), dtype=torch.uint8, device=d), sh.layout,
                                  torch.zeros(cap, dtype=torch.int32, device=d),
                                  torch.zeros(cap, dtype=torch.int32, device=d),
                                  torch.zeros(cap, dtype=torch.uint8, device=d),
                                  torch.zeros(cap, dtype=torch.uint8, device=d), 0, {})
        sh.chunk_nbytes = sh.layout.sc * sh.layout.vc * 2
        sh.n_chunks = sh.res.G.numel() // sh.chunk_nbytes
        sh.dst = torch.empty(sh.n_chunks * (sh.chunk_nbytes + 32), dtype=torch.uint8, device=d)
        sh.off = torch.zeros(sh.n_chunks + 1, dtype=torch.int64, device=d)
        sh.total = 0
        sh.cmp_done = None
        sh.cursor = torch.zeros(1, dtype=torch.int64, device=d)
        sh.pending = dev.PendingEncode()
        sh.max_lines = V + 64
        shards.append(sh)
    return shards


def one_step(ctx, shards, S, blocksize, streams=None, lookahead=1):
    """encode + pad + compress of every shard.  With two streams the (issue-bound) LZ4 kernel of shard k
    overlaps the (HBM-bound) index/encode kernels of shard k+1 — the same software pipeline the streaming
    converter uses; every kernel still runs once per shard per step."""
    def encode(sh):
        # asynchronous chain: the append position and every count stay on the device (hhgt_encode_text_async), so
        # the host never waits inside a step and the encode stream runs ahead of the compress stream by itself
        sh.cursor.zero_()
        ctx.encode_text_async(sh.text, S, sh.res, sh.cursor, max_lines=sh.max_lines, region=sh.contig, pending=sh.pending)
        ctx.pad_tail_cursor(sh.res, sh.cursor)

    if streams is None:
        for sh in shards:
            encode(sh)
            ctx.compress(sh.res.G, sh.chunk_nbytes, typesize=2, blocksize=blocksize, fmt=dev.BLOSC2,
                         dst=sh.dst, chunk_off=sh.off, sync=False)
        return
    s_enc, s_cmp = streams

    def enc(sh):
        with torch.cuda.stream(s_enc):
            if sh.cmp_done is not None:
                s_enc.wait_event(sh.cmp_done)          # G of this shard is free again
            encode(sh)
            sh.ready = s_enc.record_event()

    def cmp_(sh):
        with torch.cuda.stream(s_cmp):
            s_cmp.wait_event(sh.ready)
            ctx.compress(sh.res.G, sh.chunk_nbytes, typesize=2, blocksize=blocksize, fmt=dev.BLOSC2,
                         dst=sh.dst, chunk_off=sh.off, sync=False)
            sh.cmp_done = s_cmp.record_event()

    # nothing blocks the host: the whole step is queued at once; `lookahead` only bounds how far the encode stream may
    # run ahead of the compress stream (G of shard k is rewritten by the next step's encode)
    n = len(shards)
    for i in range(min(lookahead, n)):
        enc(shards[i])
    for k in range(n):
        cmp_(shards[k])
        if k + lookahead < n:
            enc(shards[k + lookahead])


def cpu_baseline(ctx, shards, S, target_s, extended=False):
    """oracle (CPU restatement of the reference path, matrix-shaped, 1 thread) on a bounded sample of
    the same workload: the first n lines of the largest shard."""
    from oracle import oracle
    sh = max(shards, key=lambda s: s.V)
    bytes_per_line = sh.nbytes / max(sh.V, 1)

    def run(n_lines):
        nb = min(sh.nbytes, int(n_lines * bytes_per_line) + 65536)
        host = sh.text[:nb].cpu().numpy()
        last_nl = int(np.flatnonzero(host == 10)[-1]) + 1
        host = host[:last_nl]
        t0 = time.perf_counter()
        o = oracle.vcf_encode(host, S, region=sh.contig, cap=n_lines + 64)
        t1 = time.perf_counter()
        G = o["G"]                       # [S, V, 2]; one sample row per Blosc block, 64 rows per chunk
        V = G.shape[1]
        cbytes = 0
        blk = V * 2
        for s0 in range(0, S, 64):
            raw = np.ascontiguousarray(G[s0:s0 + 64]).reshape(-1).view(np.uint8)
            cbytes += oracle.blosc_compress(raw, 2, min(blk, 65536 - (65536 % 2))).size
        t2 = time.perf_counter()
        return V, t1 - t0, t2 - t1, cbytes, G.size

    V0, te, tc, _, _ = run(4000)
    per_line = (te + tc) / max(V0, 1)
    n = int(min(sh.V, max(4000, target_s / max(per_line, 1e-9))))
    V, te, tc, cb, raw = run(n)
    out = {
        "value": V / (te + tc), "unit": "variants/s", "cores": 1, "kind": "port",
        "sample": f"first {V} variants of {sh.contig} ({S} samples): oracle encode {te:.2f}s + shuffle/LZ4/Blosc2 {tc:.2f}s, ratio {raw / max(cb, 1):.2f}",
    }
    if extended:
        out.update(cpu_baseline_extended(oracle, sh, S, bytes_per_line, n))
    return out


def cpu_baseline_extended(oracle, sh, S, bytes_per_line, n_lines):
    """SURVEY.md §8d's other two baseline shapes (--cpu-extended; not part of the default run):
    matrix-shaped on all host cores (the oracle is plain C behind ctypes: threads run it in parallel) and
    reference-shaped — one call per sample that rescans the whole text, as /root/reference/cpp/parse_vcf.cpp:30-71
    is driven by vcf_to_h5.py:96-101 — timed for a few samples and scaled to S calls."""
    from concurrent.futures import ThreadPoolExecutor
    nb = min(sh.nbytes, int(n_lines * bytes_per_line) + 65536)
    host = sh.text[:nb].cpu().numpy()
    host = host[:int(np.flatnonzero(host == 10)[-1]) + 1]
    T = min(os.cpu_count() or 1, 64)
    nl = np.flatnonzero(host == 10)
    cuts = [0] + [int(nl[min(len(nl) - 1, (len(nl) * (i + 1)) // T - 1)]) + 1 for i in range(T)]
    pieces = [host[cuts[i]:cuts[i + 1]] for i in range(T) if cuts[i + 1] > cuts[i]]

    def work(piece):
        o = oracle.vcf_encode(piece, S, region=sh.contig)
        G = o["G"]
        for s0 in range(0, S, 64):
            raw = np.ascontiguousarray(G[s0:s0 + 64]).reshape(-1).view(np.uint8)
            oracle.blosc_compress(raw, 2, min(G.shape[1] * 2, 65536))
        return G.shape[1]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        Vall = sum(ex.map(work, pieces))
    t_all = time.perf_counter() - t0
    # reference-shaped: 3 single-sample calls over a 20 k-line slice
    k = min(len(nl), 20000)
    sl = host[:int(nl[k - 1]) + 1]
    t0 = time.perf_counter()
    for si in (0, S // 2, S - 1):
        oracle.vcf_load_sample(sl, S, si, region=sh.contig)
    t_call = (time.perf_counter() - t0) / 3
    return {
        "all_cores": {"value": Vall / t_all, "unit": "variants/s", "cores": T, "sample": f"{Vall} variants in {len(pieces)} line-aligned pieces"},
        "reference_shaped": {"value": k / (t_call * S), "unit": "variants/s", "cores": 1,
                             "sample": f"one rescan per sample ({t_call * 1e3:.1f} ms per call over {k} lines) x {S} samples, parse only"},
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(local_rank)
    ctx = dev.Context(local_rank)
    ctx.set_clevel(args.clevel)
    S = args.samples
    shards = build_shards(ctx, args, rank, world)
    my_variants = sum(sh.V for sh in shards)
    text_bytes = sum(sh.nbytes for sh in shards)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the HBM-bound index/encode kernels go on a high-priority stream so that the many small LZ4 workgroups
    # of the previous shard do not starve them of LDS
    streams = None if args.no_overlap else (torch.cuda.Stream(priority=-1), torch.cuda.Stream())
    for _ in range(args.warmup):
        one_step(ctx, shards, S, args.blocksize, streams, args.lookahead)
    barrier()
    ctx.profile(True)
    # one un-timed single-stream pass: clean per-stage device times (with two streams the event pairs of the
    # non-dominant stages also contain the time they spend queued behind the other stream's kernels)
    ctx.profile_reset()
    one_step(ctx, shards, S, args.blocksize, None)
    torch.cuda.synchronize()
    stages_serial = ctx.profile_read()
    barrier()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(ctx, shards, S, args.blocksize, streams, args.lookahead)
    barrier()
    dt = time.perf_counter() - t0
    stages = ctx.profile_read()
    ctx.profile(False)

    dt_max, total_variants = sharding.reduce_job(dist if use_dist else None, dt, my_variants, device="cuda")

    # sizes for the roofline (algorithmic bytes), this rank
    comp_bytes = sum(int(sh.off[-1].item()) for sh in shards)
    for sh in shards:
        rec = sh.pending.wait()                                             # raises on malformed text / capacity
        assert rec.cursor_after == sh.V, (sh.contig, rec.cursor_after, sh.V)
    g_bytes = sum(sh.V * 2 * S for sh in shards)                             # V' * 2S (every synthetic record is kept)
    alg = {
        "index": text_bytes, "fixed": 0, "encode": sum(sh.V * 4 * S for sh in shards) + g_bytes,
        "lz4": g_bytes + comp_bytes, "frame": 2 * comp_bytes,
    }
    # dominant kernel stage = largest share of device time
    dom = max((k for k in stages_serial if k in alg), key=lambda k: stages_serial[k]["ms"])
    dom_ms_per_launch = stages[dom]["ms"] / max(stages[dom]["launches"], 1)
    dom_bytes_per_launch = alg[dom] / max(len(shards), 1)                     # one launch per shard per step
    achieved = dom_bytes_per_launch / (dom_ms_per_launch * 1e-3) / 1e9
    # HBM traffic of the dominant kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected
    # in separate rocprofv3 runs and corrected as MI355X_MICROARCH.md prescribes), scaled to this launch size
    traffic = None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))   # newest build last (r01, r01b, r02 ...)
    if pmcs:
        try:
            per_variant = json.load(open(pmcs[-1])).get(dom, {}).get("hbm_bytes_per_variant")
            if per_variant and S == 2504:
                traffic = per_variant * my_variants / max(len(shards), 1)
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": {"lz4": "k_lz4_blocks", "encode": "k_encode_tiles", "index": "k_index_newlines",
                                       "frame": "k_frame_write"}.get(dom, dom),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_launch": dom_bytes_per_launch, "ms_per_launch": dom_ms_per_launch}
    if dom == "lz4":
        # the contract's roofline is HBM or MFMA; this kernel is bound by neither
        roof["limiter"] = "instruction issue: VALU and scalar unit ~90 % busy per PMC (profiles/*_pmc_lz4_sq.csv), HBM idle"

    out = {
        "metric": "variants/sec encode+compress, 3M-variant x 2.5k-sample VCF",
        "value": total_variants * args.steps / dt_max, "unit": "variants/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"1000G-style {args.variants} variants x {S} samples, 22 per-chromosome shards "
                               f"(BASELINE configs[2]), biallelic phased GT-only text resident in HBM",
                   "variants_per_gpu": my_variants, "samples": S, "text_bytes_per_gpu": text_bytes,
                   "chunk": f"64 samples x {args.vc} variants x 2 int8, Blosc2 block {args.blocksize} B, typesize 2 (byte-shuffle), LZ4 clevel {args.clevel}",
                   "compression_ratio": g_bytes / max(comp_bytes, 1),
                   "parallelism": f"per-chromosome shards x{world}, no collective",
                   "streams": 1 if args.no_overlap else 2},
        "roofline": roof,
        "stages_ms_per_step": {k: v["ms"] for k, v in stages_serial.items()},
        "stages_ms_per_step_timed_region": {k: v["ms"] / args.steps for k, v in stages.items()},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctx, shards, S, args.cpu_seconds, args.cpu_extended)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
