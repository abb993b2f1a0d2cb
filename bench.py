#!/usr/bin/env python3
"""bench.py — variants/s, encode + compress, on synthetic 1000G-style VCF text resident in HBM.

One "step" = one pass of the hot path over the whole workload: for each of the 22 per-chromosome
shards (3 000 000 variants x 2504 samples, BASELINE.json configs[2] — the configuration the metric
is quoted on; ~30 GB of text, fits one MI355X):
    hhgt_encode_text_planes_async (line index -> fixed columns/filter -> GT tiles)  ->  the genotype matrix as two
                                   bits per allele (include/hhgt.h "Bit-plane form"; --intermediate int8: the int8 matrix)
    hhgt_pad_tail_planes_cursor
    hhgt_compress_planes (byte-shuffle + LZ4 -> Blosc2-framed chunks that decode to the int8 matrix)
Inputs (raw VCF text) are generated ON the GPU (csrc/synth.hip) before the timed region; nothing in the
step waits on the host.  `value` is this kernel-only leg.  Outside the timed region and
outside `value`, rank 0 at N = 1 also reports
    correctness  every shard's chunks decoded on the GPU and compared with G, sampled variants against the
                 generator's rule, two chunks decoded by the CPU oracle — a mismatch fails the run (rc 1)
    host_fed     §8d (ii): the same text in pinned host memory through the ingest engine (PCIe-inclusive)
    e2e          §8d (iii): BGZF level-6 shard files -> framed chunks on the host, host inflater (the north star's
                 design) and device inflater
    cpu_baseline the oracle on the host cores: matrix-shaped on 1 core and on all granted cores, and
                 reference-shaped (one rescan of the text per sample)

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks
time); shards are independent, there is no data-path collective.  `--gpus N` without a launcher starts the
N ranks itself (before this process touches a GPU); under torchrun the given environment is used.  Default
"strong": the 22 shards of ONE cohort are dealt to the ranks longest-processing-time-first, as the north star's
per-chromosome sharding does (`value` = 3 M variants x steps over the slowest rank's time); the trivially linear
weak form (every rank its own cohort) is measured in the same run and reported as the secondary field `weak`.
The end-to-end legs run on EVERY rank at once, each over its own share of the 22 BGZF files with its share of the
host CPUs (sharding.pin_rank: the CPUs of its GPU's NUMA node, divided among the ranks), and are reported as the
aggregate.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variants", type=int, default=3_000_000)
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): the 22 shards of ONE cohort are dealt to the ranks, longest first — the north star's "
                         "per-chromosome sharding; weak: every rank encodes a cohort of its own")
    ap.add_argument("--no-weak", action="store_true", help="N > 1, --scaling strong: skip the secondary weak-scaling measurement")
    ap.add_argument("--vc", type=int, default=8192, help="variants per chunk (chunk = 64 x vc x 2 bytes)")
    ap.add_argument("--blocksize", type=int, default=8192, help="Blosc2 block bytes")
    ap.add_argument("--clevel", type=int, default=5, help="codec level (reference: 5); 1-2 = run-only fast mode")
    ap.add_argument("--intermediate", choices=["planes", "int8"], default="planes",
                    help="what the encoder hands the compressor: bit planes (2 bits per allele) or the int8 matrix (round 2's step)")
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no encode/compress overlap")
    ap.add_argument("--lookahead", type=int, default=1, help="shards the encode stream runs ahead of the compress stream")
    ap.add_argument("--dist-backend", default="", help="nccl (= RCCL; default) or gloo (ranks sharing one GPU in rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU work of each baseline shape")
    ap.add_argument("--no-legs", action="store_true", help="skip the host-fed / end-to-end legs")
    ap.add_argument("--legs-chroms", default="all", help="shards the end-to-end legs run on (default: every shard of the rank)")
    ap.add_argument("--fed-chroms", default="1,2,3,4", help="shards of the host-fed leg (their text is pinned in host memory)")
    ap.add_argument("--no-check", action="store_true", help="skip the correctness gate")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C4 legs")
    ap.add_argument("--only-config", default="", help="development: run only this config leg (C2 or C4[:variants]) and print it")
    ap.add_argument("--frame-stream", action="store_true",
                    help="development: the framing half of a compress call on a third stream (hhgt_set_frame_stream); measured "
                         "in round 4: 22.9 against 22.1 ms per step — not the default")
    ap.add_argument("--lz4-priority", action="store_true", help="development: the compress stream gets the high priority")
    ap.add_argument("--cu-split", default="", help="development: 'E,C' = the encode stream may use the first E CUs of the mask "
                                                   "order, the compress stream the last C (hipExtStreamCreateWithCUMask)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# multi-GPU launch without an outside launcher
# ---------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N and no WORLD_SIZE: start the N ranks here.  This process has not touched a GPU (it imports nothing
    that initialises HIP), so the children are ordinary child processes, not re-executions of a GPU process."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


class Shard:
    pass


def build_shards(ctx, args, rank, world):
    import torch
    from haplohyped_varawareml_amd import device as dev, sharding, synth
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
        sh.contig, sh.V, sh.text, sh.nbytes, sh.seed, sh.tab, sh.chrom = contig, V, text, nbytes, seed, tab, ci + 1
        sh.layout = dev.make_layout(S, V, vc=args.vc)
        cap = sh.layout.v_capacity
        d = ctx.device
        planes = args.intermediate == "planes"
        # planes: the int8 matrix does not exist during the step (synthetic biallelic text has no calls beyond 0 / 1 /
        # missing, so nothing needs a byte in G); the gate expands the planes afterwards
        G = None if planes else torch.zeros(dev.layout_bytes(sh.layout), dtype=torch.uint8, device=d)
        P = torch.zeros(dev.planes_bytes(sh.layout), dtype=torch.uint8, device=d) if planes else None
        sh.res = dev.EncodeResult(G, sh.layout,
                                  torch.zeros(cap, dtype=torch.int32, device=d),
                                  torch.zeros(cap, dtype=torch.int32, device=d),
                                  torch.zeros(cap, dtype=torch.uint8, device=d),
                                  torch.zeros(cap, dtype=torch.uint8, device=d), 0, {}, [], P)
        sh.chunk_nbytes = sh.layout.sc * sh.layout.vc * 2
        sh.n_chunks = dev.layout_bytes(sh.layout) // sh.chunk_nbytes
        sh.dst = torch.empty(sh.n_chunks * (sh.chunk_nbytes + 32), dtype=torch.uint8, device=d)
        sh.off = torch.zeros(sh.n_chunks + 1, dtype=torch.int64, device=d)
        sh.cmp_done = None
        sh.cursor = torch.zeros(1, dtype=torch.int64, device=d)
        sh.pending = dev.PendingEncode()
        sh.max_lines = V + 64
        shards.append(sh)
    return shards


def masked_streams(n_enc, n_cmp, n_cu=256):
    """two HIP streams restricted to disjoint (or overlapping) CU sets, wrapped for torch"""
    import ctypes as C
    import torch
    hip = C.CDLL("libamdhip64.so")
    out = []
    for lo, hi in ((0, n_enc), (n_cu - n_cmp, n_cu)):
        words = (C.c_uint32 * (n_cu // 32))()
        for i in range(lo, hi):
            words[i // 32] |= 1 << (i % 32)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), n_cu // 32, words)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(st.value))
    return tuple(out)


def one_step(ctx, shards, S, blocksize, streams=None, lookahead=1):
    """encode + pad + compress of every shard, queued without any host wait (asynchronous chain: the append position
    and every count stay on the device).  With two streams the (issue-bound) LZ4 kernel of shard k overlaps the
    (HBM-bound) index/encode kernels of shard k+1 — the same software pipeline the ingest engine uses; every kernel
    still runs once per shard per step.  (Round 4, measured and removed: odd shards encoded through a second context on a
    second high-priority stream, so that one shard's small latency-bound kernels run beside the other's HBM-bound encode
    kernel: 23.6 against 23.1 ms per step; with a lookahead of two 23.6; with the framing on a third stream 22.1.)"""
    import torch
    from haplohyped_varawareml_amd import device as dev

    def encode(sh):
        c = ctx
        sh.cursor.zero_()
        if sh.res.P is not None:
            c.encode_text_planes_async(sh.text, S, sh.res, sh.cursor, max_lines=sh.max_lines, region=sh.contig, pending=sh.pending)
            c.pad_tail_planes_cursor(sh.res, sh.cursor)
        else:
            c.encode_text_async(sh.text, S, sh.res, sh.cursor, max_lines=sh.max_lines, region=sh.contig, pending=sh.pending)
            c.pad_tail_cursor(sh.res, sh.cursor)

    def compress(sh):
        if sh.res.P is not None:
            ctx.compress_planes(sh.res, fmt=dev.BLOSC2, dst=sh.dst, chunk_off=sh.off, sync=False)
        else:
            ctx.compress(sh.res.G, sh.chunk_nbytes, typesize=2, blocksize=blocksize, fmt=dev.BLOSC2, dst=sh.dst,
                         chunk_off=sh.off, sync=False)

    if streams is None:
        for sh in shards:
            encode(sh)
            compress(sh)
        return
    s_enc, s_cmp = streams[:2]
    # frame stream (hhgt_set_frame_stream): the chunks are complete in ITS order
    s_done = streams[2] if len(streams) > 2 and getattr(ctx, "frame_on", False) else s_cmp

    def enc(sh):
        with torch.cuda.stream(s_enc):
            if sh.cmp_done is not None:
                s_enc.wait_event(sh.cmp_done)          # G of this shard is free again
            encode(sh)
            sh.ready = s_enc.record_event()

    def cmp_(sh):
        with torch.cuda.stream(s_cmp):
            s_cmp.wait_event(sh.ready)
            compress(sh)
            sh.cmp_done = s_done.record_event()

    n = len(shards)
    for i in range(min(lookahead, n)):
        enc(shards[i])
    for k in range(n):
        cmp_(shards[k])
        if k + lookahead < n:
            enc(shards[k + lookahead])


# ---------------------------------------------------------------------------------------------------------------
# correctness gate (outside the timed region)
# ---------------------------------------------------------------------------------------------------------------
def correctness_gate(ctx, shards, S, blocksize):
    """What the timed steps left in HBM is checked; any mismatch raises (the run exits non-zero).  Per shard:
    (0) planes step: the planes are expanded to the int8 matrix G they stand for, and the same text is encoded once more
        by the int8 kernel (k_encode_tiles) — the two must agree byte for byte;
    (1) the framed chunks are decoded by the GPU decoder and compared with G byte for byte;
    (2) 64 sampled variants of G against the generator's rule (synth.genotype_bits) — the encode side;
    (3) one chunk (a different position in every shard) is decoded by the CPU oracle and compared with G."""
    import numpy as np
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    from oracle import oracle
    rep = dict(shards=len(shards), chunks_decoded_gpu=0, variants_sampled=0, chunks_decoded_oracle=0, shards_cross_checked_int8=0)
    rng = np.random.default_rng(12345)
    for si, sh in enumerate(shards):
        rec = sh.pending.wait()                      # raises on malformed text / capacity
        if rec.cursor_after != sh.V or rec.stats.n_kept != sh.V:
            raise AssertionError(f"{sh.contig}: kept {rec.stats.n_kept} of {sh.V} records")
        if rec.stats.n_general_lines:
            raise AssertionError(f"{sh.contig}: {rec.stats.n_general_lines} fixed-width lines took the variable-width path")
        lay = sh.layout
        if sh.res.P is not None:
            if rec.reserved:
                raise AssertionError(f"{sh.contig}: {rec.reserved} calls beyond 0 / 1 / missing in biallelic text")
            G = ctx.planes_expand(sh.res)
            z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
            r8 = dev.EncodeResult(z(G.numel(), torch.uint8), lay, z(lay.v_capacity, torch.int32), None, z(lay.v_capacity, torch.uint8),
                                  z(lay.v_capacity, torch.uint8), 0, {})
            c8 = z(1, torch.int64)
            ctx.encode_text_async(sh.text, S, r8, c8, max_lines=sh.max_lines, region=sh.contig).wait()
            ctx.pad_tail_cursor(r8, c8)
            if not torch.equal(r8.G, G) or not torch.equal(r8.start, sh.res.start):
                raise AssertionError(f"{sh.contig}: the expanded planes differ from the int8 encoder's matrix")
            rep["shards_cross_checked_int8"] += 1
            del r8
        else:
            G = sh.res.G
        back, bad = ctx.decompress(sh.dst, sh.off, sh.n_chunks, sh.chunk_nbytes, typesize=2, blocksize=blocksize)
        if bad or not torch.equal(back, G):
            raise AssertionError(f"{sh.contig}: decoded chunks differ from the genotype matrix ({bad} chunks flagged)")
        rep["chunks_decoded_gpu"] += sh.n_chunks
        del back
        vs = np.unique(np.concatenate([[0, sh.V - 1], rng.integers(0, sh.V, 62)]))
        g = G.view(torch.int8).view(lay.v_capacity // lay.vc, -(-S // lay.sc), lay.sc, lay.vc, 2)
        for v in vs:
            want = synth.genotype_bits(sh.seed, int(v), 1, S, sh.tab["thr"][v:v + 1])[0]          # [S, 2]
            got = g[int(v) // lay.vc, :, :, int(v) % lay.vc, :].reshape(-1, 2)[:S].cpu().numpy()
            if not np.array_equal(got, want.astype(np.int8)):
                raise AssertionError(f"{sh.contig}: variant {v} differs from the generator's genotypes")
        rep["variants_sampled"] += len(vs)
        k = (si * 37) % sh.n_chunks if si else 0
        if si == len(shards) - 1:
            k = sh.n_chunks - 1
        off = sh.off[k:k + 2].cpu().numpy()
        chunk = sh.dst[int(off[0]):int(off[1])].cpu().numpy()
        raw = G[k * sh.chunk_nbytes:(k + 1) * sh.chunk_nbytes].cpu().numpy()
        if not np.array_equal(oracle.blosc_decompress(chunk), raw):
            raise AssertionError(f"{sh.contig}: chunk {k} does not decode (oracle) to the matrix bytes")
        rep["chunks_decoded_oracle"] += 1
        del G
    rep["ok"] = True
    return rep


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): the oracle on the host cores
# ---------------------------------------------------------------------------------------------------------------
def effective_cpus():
    import ctypes
    from haplohyped_varawareml_amd import _lib
    L = _lib.load()
    L.hhgt_effective_cpus.restype = ctypes.c_int
    return int(L.hhgt_effective_cpus())


def cpu_baseline(shards, S, target_s):
    """SURVEY.md §8d's three shapes, each bounded to about target_s seconds of CPU work on a sample of the same
    workload (the first n lines of the largest shard):
      value / port      matrix-shaped, 1 core: one pass over the text encodes every sample, then shuffle + LZ4 + framing
      all_cores         the same on every CPU the process is granted (line-aligned pieces, one thread each: the
                        oracle is plain C behind ctypes, the GIL is released)
      reference_shaped  one call per sample that rescans the whole text, as /root/reference/cpp/parse_vcf.cpp:30-71 is
                        driven by vcf_to_h5.py:96-101 — timed for a few samples, scaled to S calls"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    sh = max(shards, key=lambda s: s.V)
    bytes_per_line = sh.nbytes / max(sh.V, 1)

    def sample(n_lines):
        nb = min(sh.nbytes, int(n_lines * bytes_per_line) + 65536)
        host = sh.text[:nb].cpu().numpy()
        return host[:int(np.flatnonzero(host == 10)[-1]) + 1]

    def work(piece):
        o = oracle.vcf_encode(piece, S, region=sh.contig)
        G = o["G"]
        V = G.shape[1]
        cb = 0
        for s0 in range(0, S, 64):
            raw = np.ascontiguousarray(G[s0:s0 + 64]).reshape(-1).view(np.uint8)
            cb += oracle.blosc_compress(raw, 2, min(V * 2, 65536 - (65536 % 2))).size
        return V, cb, G.size

    t0 = time.perf_counter()
    V0, _, _ = work(sample(2000))
    per_line = (time.perf_counter() - t0) / max(V0, 1)
    n1 = int(min(sh.V, max(2000, target_s / max(per_line, 1e-9))))
    host = sample(n1)
    t0 = time.perf_counter()
    V1, cb, raw = work(host)
    t_one = time.perf_counter() - t0
    cores = effective_cpus()
    out = {"value": V1 / t_one, "unit": "variants/s", "cores": 1, "kind": "port",
           "sample": f"first {V1} variants of {sh.contig} ({S} samples): oracle encode + shuffle/LZ4/Blosc2 on one core in "
                     f"{t_one:.2f}s, ratio {raw / max(cb, 1):.2f}"}
    # all cores: cores x the single-core sample, cut at line ends
    host = sample(min(sh.V, n1 * cores))
    nl = np.flatnonzero(host == 10)
    cuts = [0] + [int(nl[min(len(nl) - 1, (len(nl) * (i + 1)) // cores - 1)]) + 1 for i in range(cores)]
    pieces = [host[cuts[i]:cuts[i + 1]] for i in range(cores) if cuts[i + 1] > cuts[i]]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        Vall = sum(r[0] for r in ex.map(work, pieces))
    t_all = time.perf_counter() - t0
    out["all_cores"] = {"value": Vall / t_all, "unit": "variants/s", "cores": cores, "kind": "port",
                        "sample": f"{Vall} variants in {len(pieces)} line-aligned pieces, one thread each, {t_all:.2f}s"}
    # reference-shaped: single-sample calls over a slice sized for ~target_s / 3 per call
    k = int(min(len(nl), max(2000, n1 // 2)))
    sl = host[:int(nl[k - 1]) + 1]
    t0 = time.perf_counter()
    for si in (0, S // 2, S - 1):
        oracle.vcf_load_sample(sl, S, si, region=sh.contig)
    t_call = (time.perf_counter() - t0) / 3
    out["reference_shaped"] = {"value": k / (t_call * S), "unit": "variants/s", "cores": 1, "kind": "port",
                               "sample": f"one rescan of the text per sample ({t_call * 1e3:.0f} ms per call over {k} lines) x {S} "
                                         f"samples, parse only (no re-pack, no codec)"}
    return out


# ---------------------------------------------------------------------------------------------------------------
# host-fed and end-to-end legs (rank 0, N = 1): the ingest engine on a bounded set of shards
# ---------------------------------------------------------------------------------------------------------------
def ingest_legs(ctx, shards, S, e2e_chroms, fed_chroms, fmt, dist, world, reduce_device, host):
    """SURVEY.md §8d (ii) and (iii) through the native ingest engine, on THIS rank's shards, all ranks at once; every pass
    is bracketed by a barrier and the slowest rank's time counts.  No best-of: the first pass (engine cold: its pinned
    staging is still to be allocated) and the second (steady) are reported separately.  Files live in /dev/shm."""
    import shutil
    import tempfile
    import torch
    from haplohyped_varawareml_amd import sharding
    from haplohyped_varawareml_amd.ingest import Columns, Ingest, InputEnd
    from haplohyped_varawareml_amd.reader import write_bgzf_native
    pick = [sh for sh in shards if e2e_chroms is None or sh.chrom in e2e_chroms]
    fed = [sh for sh in shards if sh.chrom in fed_chroms]
    if world > 1 and not fed:       # a rank without any of the named shards feeds its own largest ones, ~9 GB at most
        acc = 0
        for sh in sorted(shards, key=lambda x: -x.nbytes):
            if acc + sh.nbytes <= 9.5e9:
                fed.append(sh)
                acc += sh.nbytes
    d = tempfile.mkdtemp(prefix="hhgt_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def agg(seconds, units):
        return sharding.reduce_job(dist, seconds, units, device=reduce_device)

    try:
        t0 = time.perf_counter()
        hosts = {}
        files = []
        for sh in pick:
            h = torch.empty(sh.nbytes, dtype=torch.uint8)
            if sh in fed:
                h = h.pin_memory()
                hosts[sh.chrom] = h
            h.copy_(sh.text)
            p = os.path.join(d, f"{sh.contig}.filtered.vcf.gz")
            write_bgzf_native(p, h.numpy(), level=6, n_threads=host["n_threads"] if world > 1 else 0)
            files.append(p)
            del h
        for sh in fed:                      # (a fed shard outside the e2e set)
            if sh.chrom not in hosts:
                hosts[sh.chrom] = torch.empty(sh.nbytes, dtype=torch.uint8).pin_memory()
                hosts[sh.chrom].copy_(sh.text)
        prep = time.perf_counter() - t0
        file_bytes = sum(os.path.getsize(p) for p in files)

        def h2d_rate():
            if not fed:
                return 0.0
            h0 = hosts[fed[0].chrom]
            dd = torch.empty(h0.numel(), dtype=torch.uint8, device=ctx.device)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                dd.copy_(h0, non_blocking=True)
            torch.cuda.synchronize()
            return 3 * h0.numel() / (time.perf_counter() - t) / 1e9

        link = h2d_rate()
        # The files above took tens of seconds of host work with the GPU idle, and a GPU that has idled takes tens of milliseconds
        # to clock up again — inside the first pass of the first leg otherwise (host_fed: 0.22 against 0.17 s, where the same leg
        # in tools/e2e_bench.py, whose preparation keeps the GPU busy, has first pass = steady pass).  ~0.1 s of device work first:
        # the legs' `first_pass` is then what a cold ENGINE costs, not a cold clock.
        if fed:
            wu = torch.empty(256 << 20, dtype=torch.uint8, device=ctx.device)
            for _ in range(40):
                wu.add_(1)
            torch.cuda.synchronize()
            del wu
        n_threads = host["n_threads"] if world > 1 else 0      # N ranks: each its share of the granted CPUs

        def run(jobs, device_inflate, want_v, passes=2):
            out = []
            # (expect_samples: the engine makes and pins its buffers at open, as pipeline.stream_files does from the first file's header)
            with Ingest(ctx, fmt=fmt, device_inflate=device_inflate, n_threads=n_threads, expect_samples=S) as ing:
                for _ in range(passes):
                    sync_all()
                    t = time.perf_counter()
                    for src, sh in jobs:
                        ing.add_file(src, sh.contig) if isinstance(src, str) else ing.add_memory(src, sh.contig)
                    framed = kept = n_end = 0
                    if jobs:
                        for ev in ing.events():
                            if isinstance(ev, Columns):
                                framed += ev.framed.size
                            elif isinstance(ev, InputEnd):
                                kept += ev.stats["n_kept"]
                                n_end += 1
                                if n_end == len(jobs):
                                    break
                    dt = time.perf_counter() - t
                    if kept != want_v:
                        raise AssertionError(f"ingest leg kept {kept} of {want_v} records")
                    dt_max, kept_all = agg(dt, kept)
                    out.append((dt_max, kept_all, framed, sharding.gather_objects(dist, round(dt, 4))))
            return out

        def leg(res, text_bytes_all, extra):
            (t1, v_all, _, r1), (t2, _, framed, r2) = res[0], res[-1]
            d = dict({"value": v_all / t2, "unit": "variants/s", "seconds": t2, "text_GBps": text_bytes_all / t2 / 1e9,
                      "first_pass": {"value": v_all / t1, "seconds": t1}, "variants": int(v_all)}, **extra)
            if world > 1:      # which rank was the slow one (value is all ranks' units over the SLOWEST rank's time)
                d["per_rank_seconds"] = r2
                d["first_pass"]["per_rank_seconds"] = r1
            return d

        _, fed_bytes_all = agg(0.0, sum(sh.nbytes for sh in fed))
        _, e2e_bytes_all = agg(0.0, sum(sh.nbytes for sh in pick))
        _, file_bytes_all = agg(0.0, file_bytes)
        _, link_all = agg(0.0, link)
        r = run([(hosts[sh.chrom], sh) for sh in fed], False, sum(sh.V for sh in fed))
        host_fed = leg(r, fed_bytes_all, {
            "pinned_h2d_GBps": link_all, "frac_of_h2d": fed_bytes_all / r[-1][0] / 1e9 / max(link_all, 1e-9),
            "ratio": sum(sh.V for sh in fed) * S * 2 / max(r[-1][2], 1),
            "sample": f"chr{','.join(str(sh.chrom) for sh in fed)} of this rank ({world} rank(s) at once): text in pinned host memory -> "
                      f"64 MiB blocks -> hipMemcpyAsync -> encode + compress -> framed chunks copied back to pinned memory; "
                      f"second (steady) pass, the first is reported beside it"})
        e2e = {"sample": f"every shard of the cohort ({len(pick)} BGZF level-6 files on this rank, {world} rank(s) at once, "
                         f"{int(e2e_bytes_all / 1e9)} GB of text in {file_bytes_all / 1e6:.0f} MB of files in /dev/shm) -> framed chunks in "
                         f"pinned host memory; second (steady) pass, the first (engine cold) beside it; no best-of",
               "file_bytes": int(file_bytes_all), "prep_seconds": prep, "host_cpus_granted": host["granted"],
               "reader_threads_per_rank": n_threads or host["n_threads"], "numa_node": host["numa_node"],
               "default_policy": "device_inflate='auto' (pipeline.stream_files, the converter): BGZF files that compress >= 2:1 "
                                 "take the device inflater — the leg below named device_inflate IS that default; host "
                                 "(device_inflate=False / HHGT_DEVICE_INFLATE=0) is the north star's design"}
        want = sum(sh.V for sh in pick)
        _, want_all = agg(0.0, want)
        # the files -> chunks legs run at the effort the file-writing paths run at (pipeline.FILE_CLEVEL = 9: twelve candidates
        # + the lazy parse; they are bound by their input, not by the coder); the kernel-only `value` stays at --clevel
        from haplohyped_varawareml_amd.pipeline import FILE_CLEVEL
        kernel_clevel = ctx.clevel
        ctx.set_clevel(int(os.environ.get("HHGT_FILE_CLEVEL", FILE_CLEVEL)))
        e2e["clevel"] = ctx.clevel

        def sum_framed(res):
            return agg(0.0, res[-1][2])[1]

        for name, mode in (("host_inflate", False), ("device_inflate", "auto")):
            r = run([(p, sh) for p, sh in zip(files, pick)], mode, want)
            text_per_variant = e2e_bytes_all / max(want_all, 1)
            ceil = ({"bound": "host-device link: the inflated text crosses it", "pinned_h2d_GBps": link_all,
                     "variants_per_s_at_link": link_all * 1e9 / text_per_variant,
                     "frac_of_link": (e2e_bytes_all / r[-1][0] / 1e9) / max(link_all, 1e-9)} if mode is False else
                    {"bound": "k_inflate_members (one wave per BGZF member, ~2 us per symbol) + first-block latency per file; "
                              "the link carries the compressed members only",
                     "file_GBps_over_link": file_bytes_all / r[-1][0] / 1e9, "pinned_h2d_GBps": link_all})
            e2e[name] = leg(r, e2e_bytes_all, {
                "ceiling": ceil, "storage": "/dev/shm (tmpfs: page cache speed, no disk)" if d.startswith("/dev/shm") else d,
                "ratio": want_all * S * 2 / max(sum_framed(r), 1),
                "file_GBps": file_bytes_all / r[-1][0] / 1e9,
                "inflater": "hhgt_reader: own DEFLATE decoder (csrc/fast_inflate.h, zlib as fallback) + PCLMUL CRC-32 on the granted host CPUs -> pinned ring -> hipMemcpyAsync"
                if mode is False else "k_inflate_members + k_crc32_members on the device (compressed members cross PCIe), chosen by the default 'auto' policy"})
        ctx.set_clevel(kernel_clevel)
        return host_fed, e2e
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------
# the other GPU configurations of BASELINE.json (configs[1] = C2, configs[3] = C4) as legs of the same line
# ---------------------------------------------------------------------------------------------------------------
def config_leg(ctx, which, variants=None, steps=3, check=True, streams=None):
    """One pass of the hot path (encode -> pad -> compress, bit-plane intermediate, text resident in HBM) over
    C2 = synthetic chr22, 50 000 variants x 1000 samples, biallelic phased (one piece of text, single stream), or
    C4 = 500 000 variants x 5000 samples with multiallelic records (dropped by the reference's isSNP filter), ./. and .|1
         calls, '/' separators and GT:DP columns; its text (11.5 GB) is rendered and encoded in pieces below 4 GiB.  With
         `streams` (the library's encode / compress pair; development, HHGT_BENCH_C4_PIPED=1) the pass is piped the way the ingest
         engine pipes a file: piece k + 1 is queued, then the result record of piece k is read and the chunk columns it completed are
         compressed on the second stream while piece k + 1 is encoded on the first (the same chunks, in the same dst) — measured
         slower than the single-stream pass the line reports.  `stages_ms`: one single-stream pass.
    Checked after the timed passes: kept counts against the generator's table, every chunk decoded on the GPU against the
    expanded planes, sampled variants against the generator's call rule, one chunk through the CPU oracle."""
    import numpy as np
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    from oracle import oracle
    d = ctx.device
    if which == "C2":
        S, V, contig, seed = 1000, variants or 50_000, "chr22", 22
        tab = synth.variant_table(seed, V, S)
        pieces = [ctx.synth_fixed(contig, tab, S, seed=seed)[0]]
        kept = np.arange(V)
    else:
        S, V, contig, seed = 5000, variants or 500_000, "chr4", 4
        tab = synth.mixed_table(seed, V, S)
        kept = np.nonzero(tab["kept"])[0]
        pieces, step_v = [], int(os.environ.get("HHGT_BENCH_C4_PIECE", "100000"))   # (development: variants per piece of text)
        for a in range(0, V, step_v):
            sub = {k: (v[a:a + step_v] if isinstance(v, np.ndarray) and len(v) == V else v) for k, v in tab.items()}
            pieces.append(ctx.synth_mixed(contig, sub, S, seed=seed, v_first=a, with_header=(a == 0))[0])
    n_kept = len(kept)
    lay = dev.make_layout(S, n_kept)
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=d)
    cap = lay.v_capacity
    res = dev.EncodeResult(None, lay, z(cap, torch.int32), None, z(cap, torch.uint8), z(cap, torch.uint8), 0, {}, [],
                           z(dev.planes_bytes(lay), torch.uint8))
    chunk_nbytes = lay.sc * lay.vc * 2
    n_chunks = dev.layout_bytes(lay) // chunk_nbytes
    n_cols = cap // lay.vc
    per_col = n_chunks // n_cols                     # chunks of one chunk column
    slot = chunk_nbytes + 32
    dst = torch.empty(n_chunks * slot, dtype=torch.uint8, device=d)
    # a batch of columns [c0, c1) frames into dst[c0 * per_col * slot ...] with its own offset table (per_col * (c1 - c0) + 1 entries
    # at off_all[c0 * per_col + batch index]); a single-stream pass is one batch
    off_all = z(n_chunks + len(pieces) + 1, torch.int64)
    cursor = z(1, torch.int64)
    pend = [dev.PendingEncode() for _ in pieces]
    piped = streams is not None and len(pieces) > 1
    batches = []                                     # (c0, c1, batch index) of the last pass

    def compress_cols(c0, c1, bi):
        ctx.compress_planes(res, col0=c0, n_cols=c1 - c0, fmt=dev.BLOSC2, dst=dst[c0 * per_col * slot:c1 * per_col * slot],
                            chunk_off=off_all[c0 * per_col + bi:c1 * per_col + bi + 1], sync=False)
        batches.append((c0, c1, bi))

    def encode(k):
        t = pieces[k]
        ctx.encode_text_planes_async(t, S, res, cursor, max_lines=t.numel() // (2 * S + 17) + 64, region=contig, pending=pend[k])

    def one_pass():
        del batches[:]
        cursor.zero_()
        for k in range(len(pieces)):
            encode(k)
        ctx.pad_tail_planes_cursor(res, cursor)
        compress_cols(0, n_cols, 0)

    def one_pass_piped():
        s_enc, s_cmp = streams[:2]
        del batches[:]
        done = 0
        main = torch.cuda.current_stream()
        s_enc.wait_stream(main)
        s_cmp.wait_stream(main)
        with torch.cuda.stream(s_enc):
            cursor.zero_()
            encode(0)
        for k in range(1, len(pieces) + 1):
            with torch.cuda.stream(s_enc):
                if k < len(pieces):
                    encode(k)                                   # queued before piece k - 1's record is read
                else:
                    ctx.pad_tail_planes_cursor(res, cursor)
                    tail = s_enc.record_event()
            rec = pend[k - 1].wait()                            # (the host waits; the device has piece k to work on)
            c1 = n_cols if k == len(pieces) else int(rec.cursor_after) // lay.vc
            if c1 > done:
                with torch.cuda.stream(s_cmp):
                    s_cmp.wait_event(tail if k == len(pieces) else pend[k - 1].event)
                    compress_cols(done, c1, k - 1)
                done = c1
        main.wait_stream(s_enc)
        main.wait_stream(s_cmp)

    run = one_pass_piped if piped else one_pass
    one_pass()
    torch.cuda.synchronize()
    ctx.profile(True)                                # stage times: one single-stream pass (events around every stage of one chain)
    ctx.profile_reset()
    one_pass()
    torch.cuda.synchronize()
    stages = ctx.profile_read()
    ctx.profile(False)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    recs = [p.wait() for p in pend]
    if not check:   # development builds whose output is not valid (timing only): --only-config ... --no-check
        return {"value": n_kept / dt, "ms_per_pass": dt * 1e3, "stages_ms": {k: v["ms"] for k, v in stages.items()},
                "compression_ratio": 0.0, "checked": "nothing (--no-check)"}
    if int(cursor.item()) != n_kept or sum(r.stats.n_records for r in recs) != V or any(r.reserved for r in recs):
        raise AssertionError(f"{which}: kept {int(cursor.item())} of {n_kept} expected, records {sum(r.stats.n_records for r in recs)} of {V}")
    G = ctx.planes_expand(res)
    comp = 0
    col_bytes = per_col * chunk_nbytes
    for c0, c1, bi in batches:                       # every batch of the LAST timed pass, against the expanded planes
        off = off_all[c0 * per_col + bi:c1 * per_col + bi + 1]
        back, bad = ctx.decompress(dst[c0 * per_col * slot:c1 * per_col * slot], off, (c1 - c0) * per_col, chunk_nbytes, typesize=2,
                                   blocksize=8192)
        if bad or not torch.equal(back, G[c0 * col_bytes:c1 * col_bytes]):
            raise AssertionError(f"{which}: decoded chunks of columns {c0}..{c1} differ from the expanded planes ({bad} chunks flagged)")
        comp += int(off[-1].item())
        del back
    if sorted(b[:2] for b in batches)[0][0] != 0 or sum(c1 - c0 for c0, c1, _ in batches) != n_cols:
        raise AssertionError(f"{which}: the batches {batches} do not cover the {n_cols} chunk columns")
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(n_kept, min(n_kept, 512), replace=False))
    g = G.view(torch.int8).view(cap // lay.vc, -(-S // lay.sc), lay.sc, lay.vc, 2)
    pk = torch.from_numpy(pick).to(d)
    got = g[pk // lay.vc, :, :, pk % lay.vc, :].reshape(len(pick), -1, 2)[:, :S].permute(1, 0, 2).cpu().numpy()
    if which == "C2":
        want = np.stack([synth.genotype_bits(seed, int(v), 1, S, tab["thr"][v:v + 1])[0] for v in pick], axis=1).astype(np.int8)
    else:
        want = synth.mixed_expected_G(seed, tab, S, kept[pick])
    if not np.array_equal(got, want):
        raise AssertionError(f"{which}: sampled variants differ from the generator's calls")
    c0, c1, bi = batches[len(batches) // 2]          # one chunk through the CPU oracle
    k = ((c0 + c1) // 2) * per_col + per_col // 2
    o2 = off_all[k + bi:k + bi + 2].cpu().numpy() + c0 * per_col * slot
    if not np.array_equal(oracle.blosc_decompress(dst[int(o2[0]):int(o2[1])].cpu().numpy()), G[k * chunk_nbytes:(k + 1) * chunk_nbytes].cpu().numpy()):
        raise AssertionError(f"{which}: chunk {k} does not decode (oracle) to the matrix bytes")
    n_missing = int((G.view(torch.int8) == -9).sum().item())
    # nonzero bytes per 4096-variant plane: what the bit-plane coders walk (more than 636: the byte-wise kernel's)
    per_plane = (G.view(-1, lay.vc // 4096, 4096, 2) != 0).sum(dim=2).reshape(-1).float()
    dense = float((per_plane > 636).float().mean().item())
    out = {"workload": ("synthetic chr22, 50 000 variants x 1000 samples, biallelic phased (BASELINE configs[1])" if which == "C2" else
                        "500 000 variants x 5000 samples, multiallelic + missing calls (BASELINE configs[3]); the reference's isSNP filter "
                        "drops the multiallelic records") if not variants else f"{which} shape at {V} variants",
           "value": V / dt, "unit": "variants/s", "ms_per_pass": dt * 1e3, "variants": V, "kept": n_kept, "samples": S,
           "text_bytes": int(sum(t.numel() for t in pieces)), "general_lines": int(sum(r.stats.n_general_lines for r in recs)),
           "missing_calls": n_missing, "compression_ratio": n_kept * 2 * S / max(comp, 1),
           "nonzero_bytes_per_plane": {"mean": float(per_plane.mean().item()), "max": float(per_plane.max().item()),
                                       "planes_left_to_the_byte_wise_coder": dense},
           "stages_ms": {k_: v["ms"] for k_, v in stages.items()},
           "checked": f"{n_chunks} chunks decoded on the GPU, {len(pick)} variants against the generator, 1 chunk through the oracle",
           "streams": 2 if piped else 1, "batches": len(batches)}
    del G, res, dst, pieces
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np  # noqa: F401
    import torch
    from haplohyped_varawareml_amd import device as dev
    from haplohyped_varawareml_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1
    n_dev = max(torch.cuda.device_count(), 1)
    shared_gpu = world > n_dev            # rehearsal: more ranks than GPUs (RCCL cannot put two ranks on one device)
    backend = args.dist_backend or ("gloo" if shared_gpu else "nccl")
    local_rank = local_rank % n_dev
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    # this rank's share of the host (N > 1: pinned to the CPUs of its GPU's NUMA node, divided among the ranks; the engine's
    # reader threads inherit the mask and get host["n_threads"] of them)
    host = sharding.pin_rank(rank, world, local_rank, devices=[r % n_dev for r in range(world)])
    ctx = dev.Context(local_rank)
    ctx.set_clevel(args.clevel)
    if args.only_config:
        name, _, nv = args.only_config.partition(":")
        # (HHGT_BENCH_C4_PIPED=1: C4's pieces piped over the stream pair — measured 13.7 against 12.5 ms: five pieces are all fill and
        # drain, and smaller ones are launch-bound: 17.8 ms with 10, 30 ms with 20)
        piped = os.environ.get("HHGT_BENCH_C4_PIPED") == "1"
        print(json.dumps(config_leg(ctx, name, int(nv) if nv else None, check=not args.no_check,
                                    streams=(ctx.create_stream("encode"), ctx.create_stream("compress")) if piped else None)))
        return
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
    if args.no_overlap:
        streams = None
    elif args.cu_split:
        streams = masked_streams(*[int(x) for x in args.cu_split.split(",")])
    elif args.lz4_priority:
        streams = (torch.cuda.Stream(), torch.cuda.Stream(priority=-1))
    else:
        # the library's stream pair: encode at high priority on every CU, compress kept off a quarter of the chip so that
        # the encode chain of the next shard finds free slots (include/hhgt.h hhgt_stream_create; HHGT_COMPRESS_CUS=0
        # gives the compress stream the whole chip)
        streams = (ctx.create_stream("encode"), ctx.create_stream("compress"))
        if args.frame_stream:
            # ... and the framing half of a compress call on a third stream, beside the LZ4 kernels of the next shard
            # (include/hhgt.h hhgt_set_frame_stream): the compress stream's critical path is the LZ4 kernels alone
            streams += (ctx.create_stream("frame"),)

    def frame_stream(on):
        if streams is not None and len(streams) > 2:
            torch.cuda.synchronize()
            ctx.set_frame_stream(streams[2] if on else None)
            ctx.frame_on = on

    frame_stream(True)
    for _ in range(args.warmup):
        one_step(ctx, shards, S, args.blocksize, streams, args.lookahead)
    barrier()
    frame_stream(False)
    ctx.profile(True)
    # one un-timed single-stream pass: clean per-stage device times (with two streams the event pairs of the
    # non-dominant stages also contain the time they spend queued behind the other stream's kernels)
    one_step(ctx, shards, S, args.blocksize, None)   # (the stream pair's warm-up leaves the default stream cold: this pass read 40 % high once)
    torch.cuda.synchronize()
    ctx.profile_reset()
    one_step(ctx, shards, S, args.blocksize, None)
    torch.cuda.synchronize()
    stages_serial = ctx.profile_read()
    frame_stream(True)
    barrier()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(ctx, shards, S, args.blocksize, streams, args.lookahead)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0      # this rank's own steps (per_rank); the job's time is the slowest rank's, behind the barrier
    barrier()
    dt = time.perf_counter() - t0
    stages = ctx.profile_read()
    ctx.profile(False)
    frame_stream(False)

    dev_for_reduce = "cuda" if backend == "nccl" else "cpu"
    dt_max, total_variants = sharding.reduce_job(dist if use_dist else None, dt, my_variants, device=dev_for_reduce)
    # N > 1: who had what and how long it took — the reduction above keeps only the slowest rank's time
    per_rank = sharding.gather_objects(dist if use_dist else None, {
        "rank": rank, "device": local_rank, "shards": [sh.contig for sh in shards], "variants": my_variants,
        "text_bytes": text_bytes, "ms_per_step": round(dt_own / args.steps * 1e3, 3),
        "stages_ms_per_step_timed_region": {k: round(v["ms"] / args.steps, 3) for k, v in stages.items()},
        "reader_threads": host["n_threads"], "numa_node": host["numa_node"], "pinned_cpus": len(host["cpus"]),
        "cpu_range": f"{min(host['cpus'])}-{max(host['cpus'])}" if host["cpus"] else ""})

    failed = None
    check = None
    if not args.no_check:
        try:
            check = correctness_gate(ctx, shards, S, args.blocksize)
        except Exception as e:      # reported in the line, and the run exits non-zero
            failed = f"{type(e).__name__}: {e}"
            check = {"ok": False, "error": failed}

    # sizes for the roofline (algorithmic bytes), this rank
    comp_bytes = sum(int(sh.off[-1].item()) for sh in shards)
    g_bytes = sum(sh.V * 2 * S for sh in shards)                             # V' * 2S (every synthetic record is kept)
    alg = {
        "index": text_bytes, "fixed": 0, "encode": sum(sh.V * 4 * S for sh in shards) + g_bytes,
        "lz4": g_bytes + comp_bytes, "frame": 2 * comp_bytes,
    }
    planes = args.intermediate == "planes"
    p_bytes = sum(sh.res.P.numel() for sh in shards) if planes else 0
    # what the kernels of this build actually move (planes step: the matrix crosses HBM as 2 bits per allele)
    moved = {"encode": sum(sh.V * 4 * S for sh in shards) + (p_bytes if planes else g_bytes),
             "lz4": (p_bytes if planes else g_bytes) + comp_bytes}
    # dominant kernel stage = largest share of device time
    dom = max((k for k in stages_serial if k in alg), key=lambda k: stages_serial[k]["ms"])
    dom_ms_per_launch = stages[dom]["ms"] / max(stages[dom]["launches"], 1)
    dom_bytes_per_launch = alg[dom] / max(len(shards), 1)                     # one launch per shard per step
    achieved = dom_bytes_per_launch / (dom_ms_per_launch * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: NOT measured in this run — per-variant bytes of the newest committed PMC
    # passes (FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 runs on a 300 k-variant workload and corrected
    # as MI355X_MICROARCH.md prescribes), scaled to this launch size
    traffic, traffic_source = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))   # newest build last (r01, r01b, r02 ...)
    if pmcs:
        try:
            per_variant = json.load(open(pmcs[-1])).get(dom, {}).get("hbm_bytes_per_variant")
            if per_variant and S == 2504:
                traffic = per_variant * my_variants / max(len(shards), 1)
                traffic_source = (f"{os.path.basename(pmcs[-1])}: {per_variant:.0f} B/variant from separate --pmc passes "
                                  f"(300 k variants), scaled to this launch; not collected in this run")
        except Exception:
            traffic = None
    # whole path against SURVEY.md §8d's definition: B = B_enc + B_cmp (unfused: V (F + 6 S) + V' 2 S (1 + 1/r)) per
    # step, over the step time, over the HBM peak
    b_whole = text_bytes + g_bytes + g_bytes + comp_bytes
    step_s = dt_max / args.steps
    moved_per_launch = moved.get(dom, alg[dom]) / max(len(shards), 1)
    roof = {"bound": "hbm", "kernel": {"lz4": "k_lz4_bitplanes_uni" if planes else "k_lz4_bitplanes", "encode": "k_encode_planes" if planes else "k_encode_tiles", "index": "k_index_hop",
                                       "frame": "k_frame_write"}.get(dom, dom),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # what this build's kernel really moves per launch (planes in, streams out) over the same launch time: the figure
            # above prices the launch at SURVEY 8d's bytes (the int8 matrix), as the contract asks
            "achieved_moved": moved_per_launch / (dom_ms_per_launch * 1e-3) / 1e9,
            "frac_moved": moved_per_launch / (dom_ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "bytes_per_launch": dom_bytes_per_launch, "ms_per_launch": dom_ms_per_launch,
            "bytes_definition": "SURVEY 8d per-variant figure x variants per launch (lz4: 2 S read + 2 S / r written; encode: 4 S + 2 S)",
            "moved_bytes_per_launch": moved_per_launch,
            "whole_path": {"bytes_per_step": b_whole, "GBps": b_whole / step_s / 1e9, "frac": b_whole / step_s / 1e9 / HBM_PEAK_GBS,
                           "definition": "SURVEY 8d: V (F + 6 S) + V' 2 S (1 + 1/r) over the step time over 8 TB/s"}}
    if dom == "lz4" and streams is not None and not args.cu_split and not args.lz4_priority:
        # the timed region runs this kernel on the library's compress stream = 3/4 of the CUs (hhgt_stream_create):
        # longer launches, shorter step.  The same kernel on the whole chip: the single-stream pass in front of the
        # timed region.
        ser = stages_serial[dom]["ms"] / max(stages_serial[dom]["launches"], 1)
        cus = os.environ.get("HHGT_COMPRESS_CUS")
        roof["cu_mask"] = ("compress stream restricted to " + (cus if cus and cus != "0" else "3/4 of the") + " CUs" if cus != "0"
                           else "none (HHGT_COMPRESS_CUS=0)")
        roof["whole_chip"] = {"ms_per_launch": ser, "achieved": dom_bytes_per_launch / (ser * 1e-3) / 1e9,
                              "frac": dom_bytes_per_launch / (ser * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "source": "un-timed single-stream pass of the same run, default stream (every CU)"}
    if dom == "lz4":
        # the contract's roofline is HBM or MFMA; this kernel is bound by neither
        sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_lz4_sq.txt")))
        mix = ""
        try:   # instruction mix per 4 KiB plane from the newest committed SQ-counter pass (a separate rocprofv3 --pmc run)
            line = next(l for l in open(sq[-1]) if l.startswith(("k_lz4_bitplanes_uni<2, false>", "k_lz4_bitplanes<2, true, false")))
            w = json.loads(line[line.index("{"):])
            mix = (f"{w['SQ_INSTS_VALU'] / 1e3:.2f} k vector + {w['SQ_INSTS_SALU'] / 1e3:.2f} k scalar + {w['SQ_INSTS_LDS'] / 1e3:.2f} k LDS "
                   f"instructions per 4 KiB plane at 7 waves per SIMD ({os.path.basename(sq[-1])}); ")
        except Exception:
            pass
        roof["bound_note"] = ("'hbm' is the contract's vocabulary (HBM or MFMA): this kernel is bound by neither — see limiter; "
                              "achieved_moved / frac_moved price the launch at the bytes it really moves")
        roof["limiter"] = ("the sum of a wave's DEPENDENT latencies (vector -> vector, ~220 LDS round trips per plane, scalar <-> vector "
                           "hand-overs, branches) at 7 waves per SIMD: " + mix + "stage time follows occupancy, not the instruction "
                           "count (measurements: DESIGN.md 3.2; round 4: a scalar-side selection walk with a third fewer vector "
                           "instructions per window was 16 % slower); HBM at a sixth of its peak under this kernel")

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
                   "shards_per_gpu": len(shards), "host_cpus": {"granted": host["granted"], "threads_per_rank": host["n_threads"],
                                                                "numa_node": host["numa_node"], "pinned_cpus": len(host["cpus"])},
                   "parallelism": f"per-chromosome shards x{world} ({args.scaling}: "
                                  + ("the 22 shards of one cohort dealt longest-first" if args.scaling == "strong" else "a cohort per rank")
                                  + "), no collective"
                                  + (f" ({backend}: {world} ranks on {n_dev} GPU: a rehearsal, not a scaling measurement)" if shared_gpu else ""),
                   "intermediate": "bit planes, 2 bits per allele (include/hhgt.h)" if planes else "int8 matrix",
                   "streams": 1 if args.no_overlap else (3 if streams is not None and len(streams) > 2 else 2),
                   "frame_stream": bool(streams is not None and len(streams) > 2),
                   "compress_stream": None if args.no_overlap else "CU mask: 3/4 of the chip (hhgt_stream_create)"},
        "roofline": roof,
        "per_rank": per_rank if world > 1 else None,
        "correctness": check,
        "stages_ms_per_step": {k: v["ms"] for k, v in stages_serial.items()},
        "stages_ms_per_step_timed_region": {k: v["ms"] / args.steps for k, v in stages.items()},
    }
    dist_or_none = dist if use_dist else None
    # what follows is collective: a rank whose gate failed takes everybody out of it
    any_failed = sharding.reduce_job(dist_or_none, 1.0 if failed else 0.0, 0.0, device=dev_for_reduce)[0] > 0
    if not any_failed and not args.no_legs:
        try:
            e2e_chroms = None if args.legs_chroms == "all" else {int(x) for x in args.legs_chroms.split(",") if x}
            fed_chroms = {int(x) for x in args.fed_chroms.split(",") if x}
            out["host_fed"], out["e2e"] = ingest_legs(ctx, shards, S, e2e_chroms, fed_chroms, dev.BLOSC2, dist_or_none, world,
                                                      dev_for_reduce, host)
        except Exception as e:
            failed = f"ingest legs: {type(e).__name__}: {e}"
            out["host_fed"] = out["e2e"] = {"error": failed}
    if not any_failed and world > 1 and args.scaling == "strong" and not args.no_weak:
        # secondary: weak scaling — every rank its own 3 M-variant cohort (trivially linear; here so that both shapes
        # come from one run)
        for sh in shards:
            sh.text = sh.res = sh.dst = None
        torch.cuda.empty_cache()
        wargs = argparse.Namespace(**dict(vars(args), scaling="weak"))
        wshards = build_shards(ctx, wargs, rank, world)
        one_step(ctx, wshards, S, args.blocksize, streams, args.lookahead)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(ctx, wshards, S, args.blocksize, streams, args.lookahead)
        barrier()
        wdt, wv = sharding.reduce_job(dist_or_none, time.perf_counter() - t0, sum(sh.V for sh in wshards), device=dev_for_reduce)
        out["weak"] = {"value": wv * args.steps / wdt, "unit": "variants/s", "ms_per_step": wdt / args.steps * 1e3,
                       "variants_per_gpu": sum(sh.V for sh in wshards), "scaling": "weak",
                       "note": "every rank encodes its own cohort (different seeds): per-GPU work fixed"}
        del wshards
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not failed:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shards, S, args.cpu_seconds)
        if not args.no_other_configs and not failed:
            # the headline workload's buffers go first: C4's text alone is 11.5 GB
            for sh in shards:
                sh.text = sh.res = sh.dst = None
            torch.cuda.empty_cache()
            out["other_configs"] = {}
            for name in ("C2", "C4"):
                try:
                    out["other_configs"][name] = config_leg(ctx, name)
                except Exception as e:
                    failed = f"{name} leg: {type(e).__name__}: {e}"
                    out["other_configs"][name] = {"error": failed}
    if rank == 0:
        out.setdefault("cpu_baseline", None)
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write(f"bench.py: FAILED: {failed}\n")
        sys.exit(1)


if __name__ == "__main__":
    main()
