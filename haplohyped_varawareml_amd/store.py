"""On-disk container for the cohort genotype matrix: a directory of Blosc2-framed chunks + index.

The reference writes one HDF5 file with S x 22 compound datasets `donor_{id}/chr_{N}/snp_data`
(/root/reference/src/haplohyped/vcf_to_h5.py:131-135,154-180), each repeating chrom/start/stop/ref/alt
for every donor.  h5py/hdf5plugin are not available in this image (SURVEY.md §7 hard part 6), so the
primary artefact is this directory; every chunk in it is exactly the byte string an HDF5 filter-32001
pipeline would store for a (64 x 8192 x 2) int8 chunk, so `write_direct_chunk` assembly is a copy.

    <store>/meta.json                       samples, donor list, geometry, codec, groups
    <store>/<group>/chunks.bin              framed chunks, vcol-major then scol
    <store>/<group>/offsets.npy             uint64[n_chunks + 1]
    <store>/<group>/start.npy ref.npy alt.npy chrom_runs.json
group = "chr_{N}" (the reference's group naming, vcf_to_h5.py:132).
"""
import json
import os

import numpy as np

# the reference's per-donor record (vcf_to_h5.py:119-127): packed, 35 bytes
SNP_DTYPE = np.dtype([("chrom", "S5"), ("start", np.uint32), ("stop", np.uint32), ("ref", "S10"),
                      ("alt", "S10"), ("phase1", np.int8), ("phase2", np.int8)])
assert SNP_DTYPE.itemsize == 35


class StoreWriter:
    def __init__(self, path, samples, sc, vc, typesize=2, cohort_name="", donor_ids=None, chunk_format="blosc2"):
        self.path = path
        os.makedirs(path, exist_ok=True)
        assert chunk_format in ("blosc1", "blosc2")
        self.meta = dict(format="hhgt-store", version=1, cohort_name=cohort_name, samples=list(samples),
                         donor_ids=list(donor_ids) if donor_ids is not None else list(samples),
                         sc=int(sc), vc=int(vc), typesize=int(typesize), blocksize=min(int(vc) * 2, 8192),
                         chunk_format=chunk_format,
                         codec=f"{chunk_format}: byte-shuffle + LZ4 block format", groups={})
        self._cur = None

    def begin_group(self, group):
        d = os.path.join(self.path, group)
        os.makedirs(d, exist_ok=True)
        self._cur = dict(name=group, dir=d, f=open(os.path.join(d, "chunks.bin"), "wb", buffering=0), offsets=[0],
                         start=[], ref=[], alt=[], runs=[], n_variants=0, raw_bytes=0)

    def add_chunks(self, data, offsets, raw_bytes):
        """data: bytes-like of concatenated framed chunks; offsets: uint64 relative offsets [k+1]"""
        from .h5file import H5Writer
        c = self._cur
        base = c["offsets"][-1]
        mv = memoryview(data).cast("B")
        n, fd = len(mv), c["f"].fileno()
        if n >= H5Writer.PAR_MIN and H5Writer.PAR_THREADS > 1:     # large batches: several pwrite threads (see H5Writer.append)
            if getattr(self, "_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(H5Writer.PAR_THREADS)
            step = -(-(-(-n // H5Writer.PAR_THREADS)) // 4096) * 4096
            for fut in [self._pool.submit(H5Writer._pwrite_all, fd, mv[o:o + step], base + o) for o in range(0, n, step)]:
                fut.result()
        else:
            H5Writer._pwrite_all(fd, mv, base)
        c["offsets"].extend(int(base + o) for o in offsets[1:])
        c["raw_bytes"] += int(raw_bytes)

    def add_variants(self, start, ref, alt):
        c = self._cur
        c["start"].append(np.asarray(start, np.uint32).copy())
        c["ref"].append(np.asarray(ref, np.uint8).copy())
        c["alt"].append(np.asarray(alt, np.uint8).copy())
        c["n_variants"] += len(start)

    def add_chrom_runs(self, runs):
        self._cur["runs"].extend([(int(a), str(b)) for a, b in runs])

    def end_group(self):
        c = self._cur
        c["f"].close()
        np.save(os.path.join(c["dir"], "offsets.npy"), np.asarray(c["offsets"], np.uint64))
        for k in ("start", "ref", "alt"):
            arr = np.concatenate(c[k]) if c[k] else np.zeros(0, np.uint32 if k == "start" else np.uint8)
            np.save(os.path.join(c["dir"], k + ".npy"), arr)
        json.dump(c["runs"], open(os.path.join(c["dir"], "chrom_runs.json"), "w"))
        S, sc, vc = len(self.meta["samples"]), self.meta["sc"], self.meta["vc"]
        # a file with no kept SNP gives a group with no chunk column
        self.meta["groups"][c["name"]] = dict(n_variants=c["n_variants"], n_vcol=-(-c["n_variants"] // vc),
                                              n_scol=-(-max(S, 1) // sc), n_chunks=len(c["offsets"]) - 1,
                                              compressed_bytes=c["offsets"][-1], raw_bytes=c["raw_bytes"])
        self._cur = None

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown()
            self._pool = None
        json.dump(self.meta, open(os.path.join(self.path, "meta.json"), "w"), indent=1)


class GenotypeStore:
    """Reader.  Opens either the working store directory or the HDF5 file the converter exports
    (`OUT/{cohort}.h5`: read natively through h5file.H5Reader — metadata and raw chunks only).  Decoding runs on the
    GPU (hhgt_decompress_chunks); there is no CPU decode path in the product."""

    def __init__(self, path, ctx=None):
        self.path = path
        self._ctx = ctx
        self._h5 = None
        if os.path.isdir(path):
            self.meta = json.load(open(os.path.join(path, "meta.json")))
        else:
            self._open_h5(path)
        self.samples = self.meta["samples"]
        self._idx = {s: i for i, s in enumerate(self.samples)}

    def _open_h5(self, path):
        from .h5file import FILTER_BLOSC, H5Reader
        r = H5Reader(path)
        root = r.group()
        if "samples" not in root:
            raise ValueError(f"{path}: no /samples dataset — not a cohort file written by vcf_to_h5")
        samples = [x.decode() for x in r.read_array("samples")]
        donors = [x.decode() for x in r.read_array("donor_ids")] if "donor_ids" in root else list(samples)
        meta = dict(format="hhgt-h5", samples=samples, donor_ids=donors, groups={}, chunk_format="blosc1")
        self._h5_info = {}
        for name in sorted(root):
            try:
                members = r.group(root[name])
            except KeyError:
                continue
            if not name.startswith("chr_") or "genotype" not in members:
                continue
            info = r.dataset(f"{name}/genotype")
            if not info["filters"] or info["filters"][0][0] != FILTER_BLOSC:
                raise ValueError(f"{path}:{name}/genotype is not a filter-32001 dataset")
            sc, vc = int(info["chunk_shape"][0]), int(info["chunk_shape"][1])
            # typesize from the filter's client data (h5file.blosc_cd_values); the Blosc block size only exists in the
            # chunk headers, so it comes from the first group that has a chunk (a group without kept SNPs has none)
            meta.update(sc=sc, vc=vc, typesize=int(info["filters"][0][1][2]))
            meta.setdefault("blocksize", min(vc * 2, 8192))
            if info["chunks"] and "_blocksize_seen" not in meta:
                first = r.read_chunk(info, (0, 0, 0))
                meta.update(blocksize=int(first[8:12].view("<u4")[0]), _blocksize_seen=True)
            meta["groups"][name] = dict(n_variants=int(info["shape"][1]), n_vcol=-(-int(info["shape"][1]) // vc),
                                        n_scol=-(-max(len(samples), 1) // sc), n_chunks=len(info["chunks"]))
            self._h5_info[name] = info
        meta.pop("_blocksize_seen", None)
        self.meta = meta
        self._h5 = r

    def close(self):
        if self._h5 is not None:
            self._h5.close()
            self._h5 = None

    def _context(self):
        if self._ctx is None:
            from .device import Context
            self._ctx = Context(0)
        return self._ctx

    def groups(self):
        return list(self.meta["groups"])

    def variants(self, group):
        if self._h5 is not None:
            r = self._h5
            runs = list(zip((int(x) for x in r.read_array(f"{group}/chrom_run_first")),
                            (x.decode() for x in r.read_array(f"{group}/chrom_run_name"))))
            return (r.read_array(f"{group}/start"), r.read_array(f"{group}/ref").view(np.uint8),
                    r.read_array(f"{group}/alt").view(np.uint8), [list(x) for x in runs])
        d = os.path.join(self.path, group)
        return (np.load(os.path.join(d, "start.npy")), np.load(os.path.join(d, "ref.npy")),
                np.load(os.path.join(d, "alt.npy")), json.load(open(os.path.join(d, "chrom_runs.json"))))

    def _chunk_row(self, group, scol):
        """framed chunks (vcol = 0 .. n_vcol-1) of sample-chunk row `scol`, as a list of uint8 arrays"""
        g = self.meta["groups"][group]
        sc, vc = self.meta["sc"], self.meta["vc"]
        if self._h5 is not None:
            info = self._h5_info[group]
            return [self._h5.read_chunk(info, (scol * sc, v * vc, 0)) for v in range(g["n_vcol"])]
        off = np.load(os.path.join(self.path, group, "offsets.npy"))
        mm = np.memmap(os.path.join(self.path, group, "chunks.bin"), dtype=np.uint8, mode="r")
        ids = [v * g["n_scol"] + scol for v in range(g["n_vcol"])]
        return [np.asarray(mm[int(off[i]):int(off[i + 1])]) for i in ids]

    def sample_row(self, group, sample):
        """int8 [n_variants, 2] for one sample: decodes the sample's chunk row on the GPU."""
        import torch
        g = self.meta["groups"][group]
        s = self._idx[sample] if isinstance(sample, str) else int(sample)
        sc, vc = self.meta["sc"], self.meta["vc"]
        scol, sin = divmod(s, sc)
        if g["n_vcol"] == 0:
            return np.zeros((0, 2), np.int8)
        parts = self._chunk_row(group, scol)
        rel = [0]
        for part in parts:
            rel.append(rel[-1] + part.size)
        ctx = self._context()
        src = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.uint8)).to(ctx.device)
        d_off = torch.tensor(rel, dtype=torch.int64, device=ctx.device)
        chunk_nbytes = sc * vc * 2
        out, bad = ctx.decompress(src, d_off, len(parts), chunk_nbytes, typesize=self.meta["typesize"],
                                  blocksize=self.meta["blocksize"])
        if bad:
            raise RuntimeError(f"{bad} corrupt chunk(s) in {group}")
        rows = out.view(torch.int8).view(len(parts), sc, vc, 2)[:, sin].reshape(-1, 2)
        return rows[: g["n_variants"]].cpu().numpy()

    def snp_records(self, group, sample):
        """the reference's per-donor compound records (vcf_to_h5.py:119-129), synthesised on demand"""
        start, ref, alt, runs = self.variants(group)
        ph = self.sample_row(group, sample)
        rec = np.zeros(len(start), dtype=SNP_DTYPE)
        bounds = [r[0] for r in runs] + [len(start)]
        for (a, name), b in zip(runs, bounds[1:]):
            rec["chrom"][a:b] = name.encode()[:5]
        rec["start"] = start
        rec["stop"] = start + 1
        rec["ref"] = ref.view("S1")
        rec["alt"] = alt.view("S1")
        rec["phase1"] = ph[:, 0]
        rec["phase2"] = ph[:, 1]
        return rec


DONOR_CHUNK_ROWS = 7488          # 8 Blosc blocks of 936 records (32 760 B = the largest multiple of 35 under 32 KiB)


def _h5_strings(xs):
    n = max([len(x.encode()) for x in xs] + [1])
    return np.array([x.encode() for x in xs], dtype=f"S{n}")


def _h5_group_datasets(w, group, meta, g, base, off, start, ref, alt, runs):
    """the datasets of group chr_{N} (see export_h5): the chunk index of /genotype over chunk bytes already in the file at
    base + off[k], and the variant tables"""
    from .h5file import FILTER_BLOSC, blosc_cd_values
    sc, vc, S = meta["sc"], meta["vc"], len(meta["samples"])
    off = np.asarray(off, np.uint64)
    ids = np.arange(len(off) - 1)
    vcol, scol = ids // max(g["n_scol"], 1), ids % max(g["n_scol"], 1)
    chunks = [((int(sci) * sc, int(vci) * vc, 0), base + int(o0), int(o1 - o0))
              for sci, vci, o0, o1 in zip(scol, vcol, off[:-1], off[1:])]
    w.add_chunked(group, "genotype", (S, g["n_variants"], 2), np.int8, (sc, vc, 2), chunks, filter_id=FILTER_BLOSC,
                  cd_values=blosc_cd_values(meta["typesize"], sc * vc * 2), filter_name=b"blosc")
    start = np.asarray(start)
    w.add_array(group, "start", start.astype(np.uint32))
    w.add_array(group, "stop", (start + 1).astype(np.uint32))
    w.add_array(group, "ref", np.asarray(ref).astype(np.uint8).view("S1"))
    w.add_array(group, "alt", np.asarray(alt).astype(np.uint8).view("S1"))
    w.add_array(group, "chrom_run_first", np.array([r[0] for r in runs], np.uint32))
    w.add_array(group, "chrom_run_name", _h5_strings([r[1] for r in runs]) if runs else np.zeros(0, "S1"))


class H5CohortWriter:
    """StoreWriter's interface (begin_group / add_chunks / add_variants / add_chrom_runs / end_group / close, .meta) writing
    straight into OUT/{cohort}.h5: the chunk bytes of a group are appended to the file as the engine hands them over, its
    chunk index and tables follow at end_group — the file export_h5 makes from a store, without the store and without the
    second copy of every chunk (round 4: the converter's 3 M x 2504 run was 1.1 s of engine + store and 1.1 s of export).
    Used by the converter when one GPU does the work and neither the store nor the per-donor datasets are asked for."""

    def __init__(self, h5_path, samples, sc, vc, typesize=2, cohort_name="", donor_ids=None):
        from .h5file import H5Writer
        self.path = h5_path
        self.meta = dict(format="hhgt-store", version=1, cohort_name=cohort_name, samples=list(samples),
                         donor_ids=list(donor_ids) if donor_ids is not None else list(samples),
                         sc=int(sc), vc=int(vc), typesize=int(typesize), blocksize=min(int(vc) * 2, 8192),
                         chunk_format="blosc1", codec="blosc1: byte-shuffle + LZ4 block format", groups={})
        self.w = H5Writer(h5_path)
        self._cur = None
        self._named = False
        self._q = self._thread = self._err = None     # the writer thread of add_chunks(..., release=...)

    def _names(self):
        # /samples and /donor_ids first, as export_h5 writes them (the sample names arrive with the first header, before the
        # first group begins): the file comes out byte-identical to the one exported from a store
        if not self._named:
            self.w.add_array("/", "samples", _h5_strings(self.meta["samples"]))
            self.w.add_array("/", "donor_ids", _h5_strings(self.meta["donor_ids"]))
            self._named = True

    def begin_group(self, group):
        self._names()
        self._cur = dict(name=group, base=None, offsets=[0], start=[], ref=[], alt=[], runs=[], n_variants=0, raw_bytes=0)

    def add_chunks(self, data, offsets, raw_bytes, release=None):
        """release (optional): `data` stays valid until release() is called — the bytes are then written by the writer thread
        while the caller goes on (pipeline.stream_files(hold_columns=True)); without it they are written before this returns"""
        c = self._cur
        if release is None:
            addr = self.w.append(data, align=8 if c["base"] is None else 1)
        else:
            self._raise_pending()
            addr = self.w.reserve(len(data), align=8 if c["base"] is None else 1)
            if self._q is None:
                import queue
                import threading
                self._q = queue.Queue()
                self._thread = threading.Thread(target=self._write_loop, name="h5-cohort-writer", daemon=True)
                self._thread.start()
            self._q.put((addr, data, release))
        if c["base"] is None:
            c["base"] = addr
        elif addr != c["base"] + c["offsets"][-1]:
            raise RuntimeError("H5CohortWriter: the chunks of a group must follow each other in the file")
        base = c["offsets"][-1]
        c["offsets"].extend(int(base + o) for o in offsets[1:])
        c["raw_bytes"] += int(raw_bytes)

    def _write_loop(self):
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                addr, data, release = item
                try:
                    if self._err is None:
                        self.w.write_at(addr, data)
                except BaseException as e:      # (kept for the caller's thread: _raise_pending)
                    self._err = e
                finally:
                    release()
            finally:
                self._q.task_done()

    def _drain(self):
        if self._q is not None:
            self._q.join()
        self._raise_pending()

    def _raise_pending(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise e

    add_variants = StoreWriter.add_variants
    add_chrom_runs = StoreWriter.add_chrom_runs

    def end_group(self):
        c = self._cur
        S, sc, vc = len(self.meta["samples"]), self.meta["sc"], self.meta["vc"]
        # (the group's index and tables go behind its chunks in the file: reserve() has fixed the chunks' places, the writer
        # thread may still be filling them — nothing below reads them)
        g = dict(n_variants=c["n_variants"], n_vcol=-(-c["n_variants"] // vc), n_scol=-(-max(S, 1) // sc),
                 n_chunks=len(c["offsets"]) - 1, compressed_bytes=c["offsets"][-1], raw_bytes=c["raw_bytes"])
        self.meta["groups"][c["name"]] = g
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        _h5_group_datasets(self.w, c["name"], self.meta, g, c["base"] if c["base"] is not None else self.w.pos, c["offsets"],
                           cat(c["start"], np.uint32), cat(c["ref"], np.uint8), cat(c["alt"], np.uint8), c["runs"])
        self._cur = None

    def close(self):
        if self.w is not None:
            try:
                self._drain()
            finally:
                if self._q is not None:
                    self._q.put(None)
                    self._thread.join()
                    self._q = self._thread = None
            self._names()
            self.w.close()
            self.w = None


def export_h5(store_path, h5_path, donor_records=False, ctx=None):
    """store directory -> one HDF5 file at the reference's output path (`OUT/{cohort}.h5`,
    /root/reference/src/haplohyped/vcf_to_h5.py:161), written natively (h5file.py; no h5py in this image):

        /samples, /donor_ids                         fixed-length strings
        /chr_{N}/genotype   int8 [S, V', 2]          chunks (sc, vc, 2), filter 32001 (Blosc): the stored chunk bytes
                                                     ARE the store's chunks, copied once in bulk
        /chr_{N}/start, stop   uint32 [V']           0-based start, stop = start + 1 (vcfpp.h:1118-1127, SNPs)
        /chr_{N}/ref, alt      S1 [V']
        /chr_{N}/chrom_run_first, chrom_run_name     CHROM value runs (first variant index, name)

    The reference's layout (S x 22 groups `donor_{id}/chr_{N}` of 35-byte compound records) is what
    GenotypeStore.snp_records / VCFH5Reader synthesise on demand; here every genotype is stored once.
    Needs Blosc-1 framed chunks (filter 32001 is hdf5-blosc / hdf5plugin.Blosc): stores written with
    chunk_format="blosc1", which is what the converter does.

    donor_records=True adds the reference's literal layout for every donor of the sample list:
        /donor_{id}/chr_{N}/snp_data   (also linked as .../genotype, the name h5_reader.py:38-40 opens) compound (35 B packed: chrom S5, start u4, stop u4, ref S10, alt S10, phase1 i1,
                                       phase2 i1 — vcf_to_h5.py:119-135), chunks of 7488 records, filter 32001 with
                                       typesize 35 (shuffle + LZ4 on the device, like every other chunk)
    That is S x 22 datasets repeating the variant table per donor (263 GB raw for 2504 donors x 3 M variants), so the
    converter only asks for it for small cohorts."""
    from .h5file import FILTER_BLOSC, H5Writer, blosc_cd_values
    meta = json.load(open(os.path.join(store_path, "meta.json")))
    if meta.get("chunk_format", "blosc2") != "blosc1":
        raise ValueError("export_h5: filter 32001 stores Blosc-1 chunks; this store holds " + meta.get("chunk_format", "blosc2"))
    sc, vc, S = meta["sc"], meta["vc"], len(meta["samples"])

    strings = _h5_strings
    with H5Writer(h5_path) as w:
        w.add_array("/", "samples", strings(meta["samples"]))
        w.add_array("/", "donor_ids", strings(meta["donor_ids"]))
        for group, g in meta["groups"].items():
            d = os.path.join(store_path, group)
            off = np.load(os.path.join(d, "offsets.npy")).astype(np.uint64)
            start = np.load(os.path.join(d, "start.npy"))
            base = w.append_file(os.path.join(d, "chunks.bin"))       # one bulk copy of all chunk bytes
            _h5_group_datasets(w, group, meta, g, base, off, start, np.load(os.path.join(d, "ref.npy")), np.load(os.path.join(d, "alt.npy")),
                               json.load(open(os.path.join(d, "chrom_runs.json"))))
        if donor_records:
            import torch
            from .device import BLOSC1
            st = GenotypeStore(store_path, ctx=ctx)
            c = st._context()
            chunk_nbytes = DONOR_CHUNK_ROWS * SNP_DTYPE.itemsize
            for donor in meta["donor_ids"]:
                if donor not in st.samples:
                    continue
                for group in meta["groups"]:
                    rec = st.snp_records(group, donor)
                    n_chunks = -(-max(len(rec), 1) // DONOR_CHUNK_ROWS)
                    padded = np.zeros(n_chunks * DONOR_CHUNK_ROWS, dtype=SNP_DTYPE)
                    padded[:len(rec)] = rec
                    src = torch.from_numpy(padded.view(np.uint8).reshape(-1)).to(c.device)
                    dst, off, total = c.compress(src, chunk_nbytes, typesize=SNP_DTYPE.itemsize, blocksize=32760, fmt=BLOSC1)
                    off = off.cpu().numpy()
                    base = w.append(dst[:total].cpu().numpy().tobytes(), align=1)
                    chunks = [((i * DONOR_CHUNK_ROWS,), base + int(off[i]), int(off[i + 1] - off[i])) for i in range(n_chunks)]
                    w.add_chunked(f"donor_{donor}/{group}", "snp_data", (len(rec),), SNP_DTYPE, (DONOR_CHUNK_ROWS,), chunks,
                                  filter_id=FILTER_BLOSC, cd_values=blosc_cd_values(SNP_DTYPE.itemsize, chunk_nbytes),
                                  filter_name=b"blosc", aliases=("genotype",))   # the name the reference's reader opens (h5_reader.py:38-40)
    return h5_path
