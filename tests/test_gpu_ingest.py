"""-m gpu: the native streaming ingest engine (include/hhgt_ingest.h) against the oracle, for every source mode
(host reader: BGZF / gzip / plain file; BGZF inflated on the device; text in host memory), with small text blocks so
that lines, chunk columns and ring slots wrap many times."""
import gzip
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from haplohyped_varawareml_amd import device as dev, synth
from haplohyped_varawareml_amd.ingest import Columns, Header, Ingest, InputEnd, Variants
from haplohyped_varawareml_amd.pipeline import stream_files
from haplohyped_varawareml_amd.reader import write_bgzf

pytestmark = pytest.mark.gpu


def tiled_expected(G, S, V, sc, vc):
    """oracle matrix -> the chunk-tiled bytes of every chunk, in the engine's order (column-major, then sample-chunk)"""
    n_vcol, n_scol = -(-V // vc), -(-S // sc)
    out = []
    for v in range(n_vcol):
        for s in range(n_scol):
            t = np.zeros((sc, vc, 2), np.int8)
            h, w = min(sc, S - s * sc), min(vc, V - v * vc)
            t[:h, :w] = G[s * sc:s * sc + h, v * vc:v * vc + w]
            out.append(t.reshape(-1).view(np.uint8))
    return out


def run_engine(ctx, jobs, **kw):
    """-> per input: dict(names, start, ref, alt, runs, chunks (decoded), stats, cols)"""
    res = {}
    with Ingest(ctx, **kw) as ing:
        for src, region in jobs:
            ing.add_file(src, region) if isinstance(src, str) else ing.add_memory(src, region)
        ing.finish()
        order = []
        for ev in ing.events():
            r = res.setdefault(ev.input, dict(start=[], ref=[], alt=[], runs=[], chunks=[], cols=[], ended=False))
            assert not r["ended"]
            order.append((ev.input, type(ev).__name__))
            if isinstance(ev, Header):
                r["header"], r["S"] = ev.header, ev.n_samples
            elif isinstance(ev, Variants):
                assert ev.first == sum(len(x) for x in r["start"])
                r["start"].append(ev.start.copy()); r["ref"].append(ev.ref.copy()); r["alt"].append(ev.alt.copy())
                r["runs"].extend(ev.runs)
            elif isinstance(ev, Columns):
                assert ev.first_col == sum(r["cols"])
                r["cols"].append(ev.n_cols)
                off = ev.chunk_off
                assert off[0] == 0 and off[-1] == ev.framed.size
                for i in range(len(off) - 1):
                    r["chunks"].append(oracle.blosc_decompress(ev.framed[int(off[i]):int(off[i + 1])]).copy())
            elif isinstance(ev, InputEnd):
                r["stats"], r["ended"] = ev.stats, True
    # inputs come out one after the other, header first
    seen = [i for i, _ in order]
    assert seen == sorted(seen)
    for i in res:
        kinds = [k for j, k in order if j == i]
        assert kinds[0] == "Header" and kinds[-1] == "InputEnd"
    return res


def check_against_oracle(r, text, S, region, sc, vc, runs=None):
    o = oracle.vcf_encode(text, S, region=region)
    V = o["n_kept"]
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    assert r["S"] == S and r["stats"]["n_kept"] == V and r["stats"]["n_samples"] == S
    assert np.array_equal(cat(r["start"], np.uint32), o["start"])
    assert np.array_equal(cat(r["ref"], np.uint8), o["ref"]) and np.array_equal(cat(r["alt"], np.uint8), o["alt"])
    want = tiled_expected(o["G"], S, V, sc, vc)
    assert len(r["chunks"]) == len(want) and sum(r["cols"]) == -(-V // vc)
    for k, (a, b) in enumerate(zip(r["chunks"], want)):
        assert np.array_equal(a, b), f"chunk {k}"
    st = r["stats"]
    for k in ("n_records", "n_drop_region", "n_drop_filter", "n_haploid_padded"):
        assert st[k] == o["stats"][k], (k, st, o["stats"])
    assert st["text_bytes"] == len(text) and st["raw_bytes"] == len(want) * sc * vc * 2
    if runs is not None:
        assert r["runs"] == runs


@pytest.mark.parametrize("kind", ["bgzf", "gzip", "plain", "device", "memory"])
def test_sources_match_oracle(ctx, tmp_path, kind):
    S, V, sc, vc = 300, 9000, 64, 512
    tab = synth.variant_table(6, V, S)
    text, _ = synth.render_fixed_numpy("chr6", tab, S, seed=6)
    p = str(tmp_path / "chr6.filtered.vcf.gz")
    if kind in ("bgzf", "device"):
        write_bgzf(p, text, level=6)
    elif kind == "gzip":
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(text)
    elif kind == "plain":
        open(p, "wb").write(text)
    src = torch.frombuffer(bytearray(text), dtype=torch.uint8).pin_memory() if kind == "memory" else p
    r = run_engine(ctx, [(src, "chr6")], sc=sc, vc=vc, fmt=dev.BLOSC1, device_inflate=(kind == "device"),
                   block_bytes=(6 << 20) if kind == "device" else (1 << 20), n_threads=4)[0]
    check_against_oracle(r, text, S, "chr6", sc, vc, runs=[(0, "chr6")])
    assert r["stats"]["n_blocks"] >= (2 if kind == "device" else 5)
    assert bool(r["stats"]["is_bgzf"]) == (kind in ("bgzf", "device")) and bool(r["stats"]["device_inflate"]) == (kind == "device")
    assert r["header"].startswith(b"##fileformat") and r["header"].endswith(b"\n") and b"#CHROM" in r["header"]


def test_several_inputs_one_engine(ctx, tmp_path, golden_dir, fixture_text, fixture_golden):
    """three files of different sample counts through one engine (slots grow between inputs), the reference's own
    fixture (plain gzip, GT:GQ:DP columns -> general path) among them, one mixed C4-style file, and a region"""
    sc, vc = 64, 256
    texts, jobs = [], []
    tab = synth.variant_table(3, 3000, 40)
    t3, _ = synth.render_fixed_numpy("chr3", tab, 40, seed=3)
    p3 = str(tmp_path / "chr3.vcf.gz")
    write_bgzf(p3, t3, level=1)
    mixed = synth.render_mixed("chr4", 1200, 130, seed=4)
    p4 = str(tmp_path / "chr4.vcf.gz")
    write_bgzf(p4, mixed)
    jobs = [(p3, "chr3:1-900000"), (os.path.join(golden_dir, "chr22.filtered.vcf.gz"), "chr22"), (p4, "chr4")]
    res = run_engine(ctx, jobs, sc=sc, vc=vc, fmt=dev.BLOSC2, block_bytes=1 << 20, n_threads=3)
    check_against_oracle(res[0], t3, 40, "chr3:1-900000", sc, vc)
    check_against_oracle(res[1], fixture_text, 3, "chr22", sc, vc, runs=[(0, "chr22")])
    check_against_oracle(res[2], mixed, 130, "chr4", sc, vc)
    assert res[1]["stats"]["n_general_lines"] == 1000 and res[2]["stats"]["n_drop_filter"] > 0


def test_stream_files_callbacks(ctx, tmp_path):
    S, V = 70, 2500
    tab = synth.variant_table(8, V, S)
    text, _ = synth.render_fixed_numpy("chr8", tab, S, seed=8)
    p = str(tmp_path / "a.vcf.gz")
    write_bgzf(p, text, level=1)
    seen = dict(h=[], v=0, c=0, raw=0, end=[])
    stats = stream_files(ctx, [(p, "chr8"), (p, "chr9")], sc=64, vc=1024, block_bytes=1 << 20,
                         on_header=lambda i, names: seen["h"].append((i, len(names), names[0])),
                         on_variants=lambda i, a, b, c: seen.__setitem__("v", seen["v"] + len(a)),
                         on_columns=lambda i, g, n, f: (seen.__setitem__("c", seen["c"] + n), seen.__setitem__("raw", seen["raw"] + g.numel())),
                         on_end=lambda i, fs: seen["end"].append((i, fs.n_kept)))
    assert seen["h"] == [(0, S, "S00001"), (1, S, "S00001")] and seen["v"] == V and seen["c"] == 3
    assert seen["end"] == [(0, V), (1, 0)] and stats[0].chrom_runs == [(0, "chr8")] and stats[1].n_drop_region == V
    assert stats[0].samples == synth.sample_names(S) and seen["raw"] == 3 * 2 * 64 * 1024 * 2


def test_errors_surface(ctx, tmp_path):
    S, V = 20, 3000
    tab = synth.variant_table(5, V, S)
    text, _ = synth.render_fixed_numpy("chr5", tab, S, seed=5)
    p = str(tmp_path / "bad.vcf.gz")
    write_bgzf(p, text, level=6)
    raw = bytearray(open(p, "rb").read())
    bsize = raw[16] | (raw[17] << 8)
    raw[bsize + 1 - 8] ^= 0x55                       # CRC field of the first member
    open(p, "wb").write(raw)
    for mode in (False, True):
        with pytest.raises(dev.HhgtError, match="CRC"):
            run_engine(ctx, [(p, "chr5")], device_inflate=mode, block_bytes=1 << 20)
    q = str(tmp_path / "mal.vcf")
    open(q, "wb").write(text[:text.rfind(b"\n", 0, 4000) + 1] + b"chr5\tnotanumber\t.\tA\tC\t.\t.\t.\tGT" + b"\t0|1" * S + b"\n")
    with pytest.raises(dev.HhgtError, match="Error parsing VCF file"):
        run_engine(ctx, [(q, "")])
    with pytest.raises(dev.HhgtError, match="cannot open"):
        run_engine(ctx, [(str(tmp_path / "nope.vcf.gz"), "")])
    empty = str(tmp_path / "empty.vcf")
    open(empty, "wb").close()
    with pytest.raises(dev.HhgtError, match="no VCF header|empty"):
        run_engine(ctx, [(empty, "")])
    # the context is usable again afterwards
    res = run_engine(ctx, [(torch.frombuffer(bytearray(text), dtype=torch.uint8), "chr5")], sc=64, vc=512)
    check_against_oracle(res[0], text, S, "chr5", 64, 512)


def test_sites_only(ctx, tmp_path):
    S, V = 30, 2000
    tab = synth.variant_table(5, V, S)
    text, _ = synth.render_fixed_numpy("chr5", tab, S, seed=5)
    res = run_engine(ctx, [(torch.frombuffer(bytearray(text), dtype=torch.uint8), "chr5")], sites_only=True, block_bytes=1 << 20)[0]
    o = oracle.vcf_encode(text, 0, region="chr5")
    assert res["S"] == 0 and not res["chunks"] and np.array_equal(np.concatenate(res["start"]), o["start"])


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("HHGT_FUZZ_SEEDS", "12")))))   # HHGT_FUZZ_SEEDS=200: a longer soak
def test_randomized_engine_matches_oracle(ctx, tmp_path, seed):
    """random sample counts (some beyond 760: the hopping index), variant counts, chunk geometries, block sizes, thread
    counts and sources; two or three inputs per engine so that the per-input state is recycled; every matrix, table and
    chunk against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    sc = int(rng.choice([16, 64, 128]))
    vc = int(rng.choice([256, 512, 4096, 8192]))
    fmt = dev.BLOSC1 if seed % 2 else dev.BLOSC2
    jobs, expect = [], []
    for k in range(int(rng.integers(2, 4))):
        S = int(rng.choice([1, 3, 17, 64, 65, 300, 801, 1500]))
        V = int(rng.integers(1, 60000 // max(S // 8, 1) + 2))
        contig = f"chr{int(rng.integers(1, 23))}"
        kind = ["fixed", "mixed"][int(rng.integers(0, 2))] if S <= 300 and V <= 3000 else "fixed"
        if kind == "fixed":
            text, _ = synth.render_fixed_numpy(contig, synth.variant_table(seed * 10 + k, V, S), S, seed=seed * 10 + k)
            text = bytes(text)
        else:
            text = synth.render_mixed(contig, V, S, seed=seed * 10 + k)
        src_kind = ["bgzf", "plain", "memory", "device"][int(rng.integers(0, 4))]
        if src_kind == "memory":
            src = torch.frombuffer(bytearray(text), dtype=torch.uint8).pin_memory()
        else:
            src = str(tmp_path / (f"in{k}.vcf" + (".gz" if src_kind in ("bgzf", "device") else "")))
            if src_kind in ("bgzf", "device"):
                write_bgzf(src, text, level=int(rng.choice([1, 6])))
            else:
                open(src, "wb").write(text)
        region = contig if rng.random() < 0.7 else ""
        jobs.append((src, region, src_kind))
        expect.append((text, S, region))
    # one engine per inflater mode (device inflate is an engine option): group the jobs
    for mode in (False, True):
        sel = [i for i, j in enumerate(jobs) if (j[2] == "device") == mode]
        if not sel:
            continue
        line_max = max(len(expect[i][0]) // max(expect[i][0].count(b"\n"), 1) for i in sel) * 3
        bb = int(max(int(rng.choice([1 << 20, 3 << 20, 8 << 20])), line_max + (1 << 16)))
        res = run_engine(ctx, [(jobs[i][0], jobs[i][1]) for i in sel], sc=sc, vc=vc, fmt=fmt, device_inflate=mode,
                         block_bytes=bb, n_threads=int(rng.integers(1, 6)))
        for n, i in enumerate(sel):
            text, S, region = expect[i]
            check_against_oracle(res[n], text, S, region, sc, vc)


@pytest.mark.parametrize("shape", ["fixed", "mixed"])
def test_planes_geometry_matches_oracle(ctx, tmp_path, shape, fixture_text):
    """default-like chunk geometry (vc a multiple of 4096, 8 KiB blocks, typesize 2): the engine keeps the ring as BIT PLANES
    (include/hhgt.h "Bit-plane form") and compresses from them — same chunks as far as any decoder can tell.  Small text
    blocks: tiles straddle the append position, ring slots are recycled, the open column is padded at the end."""
    S, V, sc, vc = 300, 21000, 64, 4096
    if shape == "fixed":
        text, _ = synth.render_fixed_numpy("chr6", synth.variant_table(6, V, S), S, seed=6)
    else:      # ./. and .|1 calls, '/' separators, GT:DP lines, dropped records
        text = bytes(ctx.synth_mixed("chr6", synth.mixed_table(6, V, S), S, seed=6)[0].cpu().numpy())
    p = str(tmp_path / "chr6.filtered.vcf.gz")
    write_bgzf(p, text, level=1)
    pf = str(tmp_path / "chr22.filtered.vcf.gz")
    with gzip.open(pf, "wb") as f:
        f.write(fixture_text)
    res = run_engine(ctx, [(p, "chr6"), (pf, "chr22")], sc=sc, vc=vc, fmt=dev.BLOSC1, block_bytes=1 << 20, n_threads=3)
    check_against_oracle(res[0], text, S, "chr6", sc, vc)
    check_against_oracle(res[1], fixture_text, 3, "chr22", sc, vc)       # GT:GQ:DP columns: bits set by the variable-width kernel
    assert res[0]["stats"]["n_blocks"] >= 5


def test_gzip_stream_that_ends_on_a_block_boundary(ctx, tmp_path):
    """a gzip stream whose text fills the reader's blocks exactly: the reader's last block is EMPTY (n = 0, last = 1).  The
    engine must still pad and frame the open chunk column and end the input (it once broke out of the loop instead:
    variant rows without genotypes, no INPUT_END)."""
    S, L = 100, 1024
    names = synth.sample_names(S)
    head = synth.header_text("chr6", names)
    pad = (-len(head)) % L
    if pad < 8:
        pad += L
    head = b"##pad=" + b"x" * (pad - 7) + b"\n" + head           # the header block is a whole number of 1 KiB lines' worth
    assert len(head) % L == 0
    n_rec = (3 << 20) // L - len(head) // L                       # header + records = exactly 3 MiB = three reader blocks
    rng = np.random.default_rng(1)
    calls = rng.integers(0, 2, (n_rec, S, 2))
    lines = []
    for i in range(n_rec):
        gt = "\t".join(f"{a}|{b}" for a, b in calls[i])
        fixed = f"chr6\t{1000000 + 7 * i}\t.\tA\tC\t.\tPASS\t"
        info = "P" * (L - len(fixed) - len("\tGT\t") - len(gt) - 1)
        lines.append(f"{fixed}{info}\tGT\t{gt}\n")
        assert len(lines[-1]) == L
    text = head + "".join(lines).encode()
    assert len(text) == 3 << 20
    p = str(tmp_path / "chr6.filtered.vcf.gz")
    with gzip.open(p, "wb", compresslevel=1) as f:
        f.write(text)
    r = run_engine(ctx, [(p, "chr6")], sc=64, vc=512, fmt=dev.BLOSC1, block_bytes=1 << 20, n_threads=2)[0]
    check_against_oracle(r, text, S, "chr6", 64, 512)
    assert r["stats"]["n_kept"] == n_rec and n_rec % 512 != 0     # the last column was open when the stream ended


def test_device_and_host_inputs_alternate_in_one_engine(ctx, tmp_path):
    """device-inflated BGZF files (many blocks: two inflate streams, the line carry on a third), a plain file, pinned memory
    and a gzip stream, in turn through ONE engine with the `auto` policy: every input's matrix, tables and chunks against
    the oracle — the text buffers and staging slots go round between the two kinds of source"""
    S, sc, vc = 200, 64, 512
    jobs, texts = [], []
    for k, kind in enumerate(["bgzf", "plain", "bgzf", "memory", "gzip", "bgzf"]):
        V = 7000 + 900 * k
        tab = synth.variant_table(30 + k, V, S)
        text, _ = synth.render_fixed_numpy(f"chr{k + 1}", tab, S, seed=30 + k)
        p = str(tmp_path / f"in{k}.vcf{'' if kind == 'plain' else '.gz'}")
        if kind == "bgzf":
            write_bgzf(p, text, level=6)
        elif kind == "gzip":
            with gzip.open(p, "wb", compresslevel=1) as f:
                f.write(text)
        elif kind == "plain":
            open(p, "wb").write(text)
        jobs.append((torch.frombuffer(bytearray(text), dtype=torch.uint8).pin_memory() if kind == "memory" else p, f"chr{k + 1}"))
        texts.append(text)
    res = run_engine(ctx, jobs, sc=sc, vc=vc, fmt=dev.BLOSC1, device_inflate="auto", block_bytes=5 << 20, n_threads=3)
    for k, text in enumerate(texts):
        check_against_oracle(res[k], text, S, f"chr{k + 1}", sc, vc, runs=[(0, f"chr{k + 1}")])
    dev_inputs = [k for k in res if res[k]["stats"]["device_inflate"]]
    assert dev_inputs == [0, 2, 5] and all(res[k]["stats"]["n_blocks"] >= 2 for k in dev_inputs)


def test_held_chunk_buffers_outlive_later_events(ctx, tmp_path):
    """hhgt_ingest_hold: the chunk bytes of a Columns event stay intact while later events are taken, until the token is
    released (from another thread, as the converter's writer thread does); decoded only then, against the oracle"""
    import threading
    S, V, sc, vc = 150, 20000, 64, 512
    tab = synth.variant_table(41, V, S)
    text, _ = synth.render_fixed_numpy("chr9", tab, S, seed=41)
    p = str(tmp_path / "chr9.filtered.vcf.gz")
    write_bgzf(p, text, level=6)
    chunks, held = [], []

    def settle(k):      # decode what was held k events ago, then give the buffers back on another thread
        while len(held) > k:
            framed, off, token = held.pop(0)
            for i in range(len(off) - 1):
                chunks.append(oracle.blosc_decompress(framed[int(off[i]):int(off[i + 1])]).copy())
            t = threading.Thread(target=ing.release, args=(token,))
            t.start()
            t.join()

    n_cols = 0
    with Ingest(ctx, sc=sc, vc=vc, fmt=dev.BLOSC1, block_bytes=1 << 20, n_threads=2) as ing:
        ing.add_file(p, "chr9")
        ing.finish()
        for ev in ing.events():
            if isinstance(ev, Columns):
                held.append((ev.framed, ev.chunk_off.copy(), ing.hold()))
                n_cols += ev.n_cols
                settle(3)          # up to three batches stay held while the next events are taken
        settle(0)
    o = oracle.vcf_encode(text, S, region="chr9")
    want = tiled_expected(o["G"], S, o["n_kept"], sc, vc)
    assert n_cols == -(-o["n_kept"] // vc) and len(chunks) == len(want)
    for k, (a, b) in enumerate(zip(chunks, want)):
        assert np.array_equal(a, b), f"chunk {k}"
