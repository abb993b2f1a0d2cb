"""CPU: the native HDF5 writer (haplohyped_varawareml_amd/h5file.py) against an independent reader — the image's
libhdf5 1.10.6 through h5py 3.3 in /opt/conda (an interpreter this test only shells out to; when it is absent the
library-backed checks skip and only the self-contained ones run)."""
import functools
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from haplohyped_varawareml_amd import h5file

CONDA_PY = "/opt/conda/bin/python3.9"
HERE = os.path.dirname(os.path.abspath(__file__))


h5_used = {"libhdf5": 0}   # runs of the independent HDF5 reader in this session (tests/conftest.py prints it)


@functools.lru_cache(maxsize=None)
def have_h5py():
    if not os.path.exists(CONDA_PY):
        return False
    try:      # bounded: a cold interpreter start may take a minute, a stuck one must not stall the suite
        return subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True, timeout=300).returncode == 0
    except subprocess.TimeoutExpired:
        return False


needs_h5py = pytest.mark.skipif(not have_h5py(), reason="no independent HDF5 library in this environment")


def h5check(path, tmp_path, *dump):
    out = str(tmp_path / "check.npz")
    h5_used["libhdf5"] += 1
    r = subprocess.run([CONDA_PY, os.path.join(HERE, "h5check.py"), path, out, *dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out, allow_pickle=False)
    return {k: z[k] for k in z.files}


def build_blosc_plugin(tmp_path):
    """-> directory holding the test-built filter-32001 decoder plugin, or None (no compiler / headers / libraries)"""
    import shutil
    src = os.path.join(HERE, "h5z_blosc_min.c")
    if not (shutil.which("gcc") and os.path.exists("/opt/conda/include/hdf5.h") and os.path.exists("/opt/conda/include/blosc.h")):
        return None
    d = tmp_path / "h5plugin"
    d.mkdir(exist_ok=True)
    r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/conda/include", src, "-o", str(d / "libh5zbloscmin.so"),
                        "-L/opt/conda/lib", "-lhdf5", "-lblosc", "-Wl,-rpath,/opt/conda/lib"], capture_output=True, text=True, timeout=300)
    return str(d) if r.returncode == 0 else None


def h5read_filtered(path, tmp_path, plugin_dir, *datasets):
    """datasets read through libhdf5's filter pipeline (the way h5py + hdf5plugin users read them)"""
    out = str(tmp_path / "filtered.npz")
    env = dict(os.environ, HDF5_PLUGIN_PATH=plugin_dir)
    r = subprocess.run([CONDA_PY, os.path.join(HERE, "h5read_filtered.py"), path, out, *datasets], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out, allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_superblock_and_alignment(tmp_path):
    p = str(tmp_path / "a.h5")
    with h5file.H5Writer(p) as w:
        w.add_array("/", "x", np.arange(5, dtype=np.uint32))
    b = open(p, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8] == 0 and b[13] == 8 and b[14] == 8
    eof = struct.unpack_from("<Q", b, 40)[0]
    assert eof == len(b)
    root_hdr = struct.unpack_from("<Q", b, 56 + 8)[0]
    assert root_hdr % 8 == 0 and b[root_hdr] == 1          # object header version 1
    with pytest.raises(ValueError):
        w2 = h5file.H5Writer(str(tmp_path / "b.h5"))
        w2.add_array("/", "x", np.zeros(1, np.uint8))
        w2.add_array("/", "x", np.zeros(1, np.uint8))


@needs_h5py
def test_groups_arrays_and_strings(tmp_path):
    p = str(tmp_path / "g.h5")
    rng = np.random.default_rng(1)
    want = {}
    with h5file.H5Writer(p) as w:
        for i in range(45):                                   # more than one symbol node (32 entries each)
            a = rng.integers(0, 1 << 31, 7 + i, dtype=np.int64).astype([np.uint32, np.int8, np.uint8, np.int64, np.uint16][i % 5])
            w.add_array("many", f"d{i:02d}", a)
            want[f"many/d{i:02d}"] = a
        s = np.array([b"NA12878", b"HG00096", b"x"], dtype="S7")
        w.add_array("/", "samples", s)
        want["samples"] = s
        w.add_array("a/b/c", "deep", np.arange(6, dtype=np.uint32).reshape(2, 3))
        want["a/b/c/deep"] = np.arange(6, dtype=np.uint32).reshape(2, 3)
        w.add_array("a", "empty", np.zeros(0, np.uint32))
        want["a/empty"] = np.zeros(0, np.uint32)
    got = h5check(p, tmp_path)
    names = dict(json.loads(str(got["|names"])))
    assert names["many"] is False and names["a/b/c"] is False and names["a/b/c/deep"] is True
    for k, a in want.items():
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k


@needs_h5py
@pytest.mark.parametrize("n_v_chunks", [1, 3, 70, 5000])        # 1 leaf; several; two levels (> 64); three levels (> 4096)
def test_chunked_dataset_index(tmp_path, n_v_chunks):
    p = str(tmp_path / "c.h5")
    S, vc = 100, 4
    V = n_v_chunks * vc - 1
    rng = np.random.default_rng(n_v_chunks)
    G = rng.integers(-9, 3, (S, V, 2), dtype=np.int8)
    chunks = []
    with h5file.H5Writer(p) as w:
        for si in range(0, S, 64):
            for vi in range(0, V, vc):
                blk = np.zeros((64, vc, 2), np.int8)
                sub = G[si:si + 64, vi:vi + vc]
                blk[:sub.shape[0], :sub.shape[1]] = sub
                chunks.append(((si, vi, 0), w.append(blk.tobytes()), blk.nbytes))
        rng.shuffle(chunks)                                   # the writer sorts
        w.add_chunked("chr_22", "genotype", (S, V, 2), np.int8, (64, vc, 2), chunks)
    got = h5check(p, tmp_path)
    meta = json.loads(str(got["chr_22/genotype|meta"]))
    assert meta["shape"] == [S, V, 2] and meta["dtype"] == "int8" and meta["chunks"] == [64, vc, 2] and meta["filters"] == []
    assert np.array_equal(got["chr_22/genotype"], G)


@needs_h5py
def test_filter_32001_message_and_raw_chunks(tmp_path):
    p = str(tmp_path / "f.h5")
    rng = np.random.default_rng(3)
    payload = {}
    with h5file.H5Writer(p) as w:
        chunks = []
        for si in (0, 64):
            for vi in (0, 8192, 16384):
                raw = rng.integers(0, 256, int(rng.integers(20, 400)), dtype=np.uint8).tobytes()
                payload[(si, vi, 0)] = raw
                chunks.append(((si, vi, 0), w.append(raw, align=1), len(raw)))
        w.add_chunked("chr_1", "genotype", (100, 20000, 2), np.int8, (64, 8192, 2), chunks, filter_id=h5file.FILTER_BLOSC,
                      cd_values=h5file.blosc_cd_values(2, 64 * 8192 * 2), filter_name=b"blosc")
    got = h5check(p, tmp_path, "chr_1/genotype")
    meta = json.loads(str(got["chr_1/genotype|meta"]))
    assert meta["n_chunks"] == 6 and meta["chunks"] == [64, 8192, 2]
    (fid, flags, cd, name), = meta["filters"]
    assert fid == 32001 and cd == [2, 2, 2, 64 * 8192 * 2, 5, 1, 1] and name == "blosc"
    for off, raw in payload.items():
        key = ",".join(str(o) for o in off)
        assert bytes(got["chr_1/genotype|chunk|" + key]) == raw and int(got["chr_1/genotype|mask|" + key]) == 0


@needs_h5py
def test_compound_records_like_the_reference(tmp_path):
    """the reference's packed 35-byte record (vcf_to_h5.py:119-127) as a contiguous and as a chunked dataset"""
    from haplohyped_varawareml_amd.store import SNP_DTYPE
    rng = np.random.default_rng(5)
    n = 1000
    rec = np.zeros(n, dtype=SNP_DTYPE)
    rec["chrom"] = b"chr22"
    rec["start"] = np.sort(rng.integers(1, 1 << 28, n)).astype(np.uint32)
    rec["stop"] = rec["start"] + 1
    rec["ref"] = rng.choice([b"A", b"C", b"G", b"T"], n)
    rec["alt"] = rng.choice([b"A", b"C", b"G", b"T"], n)
    rec["phase1"] = rng.integers(-9, 2, n)
    rec["phase2"] = rng.integers(0, 2, n)
    p = str(tmp_path / "r.h5")
    rows = 256
    with h5file.H5Writer(p) as w:
        w.add_array("donor_X/chr_22", "flat", rec)
        chunks = []
        for i in range(0, n, rows):
            blk = np.zeros(rows, dtype=SNP_DTYPE)
            blk[:len(rec[i:i + rows])] = rec[i:i + rows]
            chunks.append(((i,), w.append(blk.tobytes(), align=1), blk.nbytes))
        w.add_chunked("donor_X/chr_22", "snp_data", (n,), SNP_DTYPE, (rows,), chunks, aliases=("genotype",))
    got = h5check(p, tmp_path)
    for name in ("donor_X/chr_22/flat", "donor_X/chr_22/snp_data", "donor_X/chr_22/genotype"):
        meta = json.loads(str(got[name + "|meta"]))
        assert meta["itemsize"] == 35 and meta["shape"] == [n]
        assert [(k, off) for k, _, off in meta["fields"]] == [(k, SNP_DTYPE.fields[k][1]) for k in SNP_DTYPE.names]
        assert [t for _, t, _ in meta["fields"]] == ["|S5", "uint32", "uint32", "|S10", "|S10", "int8", "int8"]
        for k in SNP_DTYPE.names:
            assert np.array_equal(got[name + "|field|" + k], rec[k]), (name, k)


def test_reader_roundtrip_own_files(tmp_path):
    """H5Reader over H5Writer output (no external library): names, arrays, compound dtype, chunk index, filters"""
    from haplohyped_varawareml_amd.store import SNP_DTYPE
    p = str(tmp_path / "rt.h5")
    rng = np.random.default_rng(9)
    rec = np.zeros(10, dtype=SNP_DTYPE)
    rec["start"] = np.arange(10)
    payload = {}
    with h5file.H5Writer(p) as w:
        w.add_array("/", "samples", np.array([b"a", b"bb"], dtype="S2"))
        w.add_array("g/h", "rec", rec)
        for i in range(40):
            w.add_array("many", f"d{i}", np.full(3, i, np.uint32))
        chunks = []
        for vi in range(0, 300 * 8, 8):
            raw = rng.integers(0, 256, 11 + vi % 7, dtype=np.uint8).tobytes()
            payload[(0, vi, 0)] = raw
            chunks.append(((0, vi, 0), w.append(raw, align=1), len(raw)))
        w.add_chunked("chr_9", "genotype", (60, 300 * 8 - 3, 2), np.int8, (64, 8, 2), chunks, filter_id=32001,
                      cd_values=h5file.blosc_cd_values(2, 64 * 8 * 2), filter_name=b"blosc", aliases=("alias",))
    r = h5file.H5Reader(p)
    assert sorted(r.group()) == ["chr_9", "g", "many", "samples"] and len(r.group(r.resolve("many"))) == 40
    assert list(r.read_array("samples")) == [b"a", b"bb"] and np.array_equal(r.read_array("many/d39"), np.full(3, 39, np.uint32))
    got = r.read_array("g/h/rec")
    assert got.dtype == SNP_DTYPE and np.array_equal(got["start"], rec["start"])
    d = r.dataset("chr_9/alias")
    assert d["shape"] == (60, 300 * 8 - 3, 2) and d["dtype"] == np.int8 and d["chunk_shape"] == (64, 8, 2)
    assert d["filters"] == [(32001, (2, 2, 2, 64 * 8 * 2, 5, 1, 1))] and len(d["chunks"]) == 300
    for off, raw in payload.items():
        assert bytes(r.read_chunk(d, off)) == raw
    r.close()


@needs_h5py
def test_reader_on_a_file_written_by_libhdf5(tmp_path):
    """the other direction: a file produced by h5py/libhdf5 (default, oldest-compatible format) read by H5Reader"""
    p = str(tmp_path / "lib.h5")
    script = (
        "import h5py, numpy as np\n"
        f"f = h5py.File({p!r}, 'w')\n"
        "f['samples'] = np.array([b'NA1', b'NA22'], dtype='S4')\n"
        "g = f.create_group('chr_1')\n"
        "g['start'] = np.arange(1000, dtype=np.uint32) * 3\n"
        "d = g.create_dataset('genotype', shape=(70, 500, 2), dtype='i1', chunks=(64, 100, 2))\n"
        "G = (np.arange(70 * 500 * 2) % 5 - 2).astype('i1').reshape(70, 500, 2)\n"
        "d[...] = G\n"
        "for i in range(30): f.create_group('many').create_dataset('x%d' % i, data=np.full(2, i, 'u2')) if i == 0 else f['many'].create_dataset('x%d' % i, data=np.full(2, i, 'u2'))\n"
        "f.close()\n")
    r = subprocess.run([CONDA_PY, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rd = h5file.H5Reader(p)
    assert sorted(rd.group()) == ["chr_1", "many", "samples"]
    assert list(rd.read_array("samples")) == [b"NA1", b"NA22"]
    assert np.array_equal(rd.read_array("chr_1/start"), np.arange(1000, dtype=np.uint32) * 3)
    assert np.array_equal(rd.read_array("many/x29"), np.full(2, 29, np.uint16))
    d = rd.dataset("chr_1/genotype")
    assert d["shape"] == (70, 500, 2) and d["chunk_shape"] == (64, 100, 2) and len(d["chunks"]) == 10 and d["filters"] == []
    G = (np.arange(70 * 500 * 2) % 5 - 2).astype("i1").reshape(70, 500, 2)
    blk = rd.read_chunk(d, (64, 400, 0)).view(np.int8).reshape(64, 100, 2)
    assert np.array_equal(blk[:6], G[64:70, 400:500])
    rd.close()


@needs_h5py
def test_libhdf5_filter_pipeline_reads_blosc_chunks(tmp_path):
    """the whole read path of a stock user: libhdf5 walks the chunk B-tree, hands each chunk to filter 32001, the filter
    (a test-built c-blosc decoder standing in for hdf5plugin) returns the bytes, h5py slices them.  Chunks here come from
    the CPU oracle's Blosc-1 encoder (the GPU encoder's output goes the same way in tests/test_gpu_pipeline.py)."""
    plugin = build_blosc_plugin(tmp_path)
    if plugin is None:
        pytest.skip("cannot build the filter plugin here")
    from oracle import oracle
    S, V, sc, vc = 70, 300, 64, 128
    rng = np.random.default_rng(11)
    G = (rng.random((S, V, 2)) < 0.05).astype(np.int8)
    G[rng.random((S, V, 2)) < 0.003] = -9
    p = str(tmp_path / "pipe.h5")
    with h5file.H5Writer(p) as w:
        chunks = []
        for si in range(0, S, sc):
            for vi in range(0, V, vc):
                blk = np.zeros((sc, vc, 2), np.int8)
                sub = G[si:si + sc, vi:vi + vc]
                blk[:sub.shape[0], :sub.shape[1]] = sub
                ck = oracle.blosc_compress(blk.reshape(-1).view(np.uint8), 2, vc * 2, oracle.BLOSC1)
                chunks.append(((si, vi, 0), w.append(ck.tobytes(), align=1), ck.size))
        w.add_chunked("chr_7", "genotype", (S, V, 2), np.int8, (sc, vc, 2), chunks, filter_id=h5file.FILTER_BLOSC,
                      cd_values=h5file.blosc_cd_values(2, sc * vc * 2), filter_name=b"blosc")
    got = h5read_filtered(p, tmp_path, plugin, "chr_7/genotype")
    assert np.array_equal(got["chr_7/genotype"], G)


def test_genotype_store_opens_an_h5_file_cpu(tmp_path):
    """GenotypeStore over a cohort .h5 (no GPU involved here: metadata, variant tables, raw chunk rows)"""
    from oracle import oracle
    from haplohyped_varawareml_amd.store import GenotypeStore
    S, V, sc, vc = 70, 300, 64, 128
    rng = np.random.default_rng(21)
    G = (rng.random((S, V, 2)) < 0.08).astype(np.int8)
    p = str(tmp_path / "cohort.h5")
    raw = {}
    with h5file.H5Writer(p) as w:
        w.add_array("/", "samples", np.array([f"S{i:03d}".encode() for i in range(S)], dtype="S4"))
        w.add_array("/", "donor_ids", np.array([b"S001", b"S069"], dtype="S4"))
        chunks = []
        for si in range(0, S, sc):
            for vi in range(0, V, vc):
                blk = np.zeros((sc, vc, 2), np.int8)
                sub = G[si:si + sc, vi:vi + vc]
                blk[:sub.shape[0], :sub.shape[1]] = sub
                ck = oracle.blosc_compress(blk.reshape(-1).view(np.uint8), 2, vc * 2, oracle.BLOSC1)
                raw[(si, vi, 0)] = ck
                chunks.append(((si, vi, 0), w.append(ck.tobytes(), align=1), ck.size))
        w.add_chunked("chr_7", "genotype", (S, V, 2), np.int8, (sc, vc, 2), chunks, filter_id=h5file.FILTER_BLOSC,
                      cd_values=h5file.blosc_cd_values(2, sc * vc * 2), filter_name=b"blosc")
        start = np.sort(rng.integers(1, 1 << 20, V)).astype(np.uint32)
        w.add_array("chr_7", "start", start)
        w.add_array("chr_7", "stop", start + 1)
        w.add_array("chr_7", "ref", np.frombuffer(b"ACGT" * (V // 4), dtype="S1"))
        w.add_array("chr_7", "alt", np.frombuffer(b"TGCA" * (V // 4), dtype="S1"))
        w.add_array("chr_7", "chrom_run_first", np.array([0], np.uint32))
        w.add_array("chr_7", "chrom_run_name", np.array([b"chr7"], dtype="S4"))
        w.add_array("other", "x", np.zeros(3, np.uint8))          # a group that is not a chromosome
    st = GenotypeStore(p)
    assert st.samples[:2] == ["S000", "S001"] and st.meta["donor_ids"] == ["S001", "S069"] and st.groups() == ["chr_7"]
    g = st.meta["groups"]["chr_7"]
    assert (st.meta["sc"], st.meta["vc"], st.meta["typesize"], st.meta["blocksize"]) == (sc, vc, 2, vc * 2)
    assert (g["n_variants"], g["n_vcol"], g["n_scol"], g["n_chunks"]) == (V, 3, 2, 6)
    s2, ref, alt, runs = st.variants("chr_7")
    assert np.array_equal(s2, start) and ref.dtype == np.uint8 and bytes(ref[:4]) == b"ACGT" and runs == [[0, "chr7"]]
    row = st._chunk_row("chr_7", 1)                                    # samples 64..127: three chunks along the variants
    assert [bytes(r) for r in row] == [raw[(64, v, 0)].tobytes() for v in (0, 128, 256)]
    # and each of them is what the oracle decodes back to the padded tile
    tile = oracle.blosc_decompress(np.frombuffer(row[2], np.uint8)).view(np.int8).reshape(sc, vc, 2)
    assert np.array_equal(tile[:S - 64, :V - 256], G[64:, 256:])
    st.close()
    with pytest.raises(ValueError):
        bad = str(tmp_path / "bad.h5")
        with h5file.H5Writer(bad) as w:
            w.add_array("/", "x", np.zeros(1, np.uint8))
        GenotypeStore(bad)


def test_cohort_writer_equals_the_export_of_a_store(tmp_path):
    """store.H5CohortWriter (the converter's direct path) against StoreWriter + export_h5 on the same batches of chunks:
    byte-identical files — with the chunks written before add_chunks returns, and with them handed to the writer thread and
    released afterwards (large batches go through the parallel pwrite path); an empty group among them"""
    import threading
    from oracle import oracle
    from haplohyped_varawareml_amd.store import GenotypeStore, H5CohortWriter, StoreWriter, export_h5
    S, sc, vc = 70, 64, 128
    rng = np.random.default_rng(5)
    samples = [f"S{i:03d}" for i in range(S)]
    groups = {"chr_3": 300, "chr_5": 0, "chr_9": 4000}

    def feed(w, release_log=None):
        w.meta["samples"] = list(samples)
        for gi, (group, V) in enumerate(groups.items()):
            g_rng = np.random.default_rng(100 + gi)
            G = (g_rng.random((S, max(V, 1), 2)) < 0.08).astype(np.int8)[:, :V]
            w.begin_group(group)
            start = np.sort(g_rng.integers(1, 1 << 24, V)).astype(np.uint32)
            n_vcol = -(-V // vc)
            for v0 in range(0, n_vcol, 7):                      # batches of up to 7 chunk columns, as the engine hands them over
                parts, offs = [], [0]
                for vci in range(v0, min(v0 + 7, n_vcol)):
                    for si in range(0, S, sc):
                        blk = np.zeros((sc, vc, 2), np.int8)
                        sub = G[si:si + sc, vci * vc:(vci + 1) * vc]
                        blk[:sub.shape[0], :sub.shape[1]] = sub
                        ck = oracle.blosc_compress(blk.reshape(-1).view(np.uint8), 2, vc * 2, oracle.BLOSC1)
                        parts.append(ck)
                        offs.append(offs[-1] + ck.size)
                data = np.concatenate(parts)
                raw = (len(offs) - 1) * sc * vc * 2
                if release_log is None:
                    w.add_chunks(data, np.asarray(offs, np.uint64), raw)
                else:
                    ev = threading.Event()
                    release_log.append(ev)
                    w.add_chunks(data, np.asarray(offs, np.uint64), raw, release=ev.set)
                a, b = v0 * vc, min((v0 + 7) * vc, V)
                w.add_variants(start[a:b], np.full(b - a, ord("A"), np.uint8), np.full(b - a, ord("G"), np.uint8))
            w.add_chrom_runs([(0, group.replace("_", ""))] if V else [])
            w.end_group()
        w.close()

    store = str(tmp_path / "c.hhgt")
    feed(StoreWriter(store, [], sc, vc, cohort_name="c", donor_ids=samples[:2], chunk_format="blosc1"))
    export_h5(store, str(tmp_path / "exported.h5"))
    feed(H5CohortWriter(str(tmp_path / "direct.h5"), [], sc, vc, cohort_name="c", donor_ids=samples[:2]))
    log = []
    old = h5file.H5Writer.PAR_MIN
    h5file.H5Writer.PAR_MIN = 4096                               # the small batches of this test through the parallel writes too
    try:
        feed(H5CohortWriter(str(tmp_path / "behind.h5"), [], sc, vc, cohort_name="c", donor_ids=samples[:2]), release_log=log)
    finally:
        h5file.H5Writer.PAR_MIN = old
    want = open(tmp_path / "exported.h5", "rb").read()
    assert open(tmp_path / "direct.h5", "rb").read() == want
    assert open(tmp_path / "behind.h5", "rb").read() == want
    assert log and all(ev.is_set() for ev in log)                # every held buffer was given back
    st = GenotypeStore(str(tmp_path / "direct.h5"))
    assert st.groups() == ["chr_3", "chr_5", "chr_9"] and st.meta["groups"]["chr_9"]["n_variants"] == 4000
    assert st.meta["groups"]["chr_5"]["n_chunks"] == 0
    st.close()
