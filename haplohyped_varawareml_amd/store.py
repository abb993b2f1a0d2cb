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
    def __init__(self, path, samples, sc, vc, typesize=2, cohort_name="", donor_ids=None):
        self.path = path
        os.makedirs(path, exist_ok=True)
        self.meta = dict(format="hhgt-store", version=1, cohort_name=cohort_name, samples=list(samples),
                         donor_ids=list(donor_ids) if donor_ids is not None else list(samples),
                         sc=int(sc), vc=int(vc), typesize=int(typesize), blocksize=min(int(vc) * 2, 8192),
                         codec="blosc2: byte-shuffle + LZ4 block format", groups={})
        self._cur = None

    def begin_group(self, group):
        d = os.path.join(self.path, group)
        os.makedirs(d, exist_ok=True)
        self._cur = dict(name=group, dir=d, f=open(os.path.join(d, "chunks.bin"), "wb"), offsets=[0],
                         start=[], ref=[], alt=[], runs=[], n_variants=0, raw_bytes=0)

    def add_chunks(self, data, offsets, raw_bytes):
        """data: bytes-like of concatenated framed chunks; offsets: uint64 relative offsets [k+1]"""
        c = self._cur
        base = c["offsets"][-1]
        c["f"].write(memoryview(data))
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
        self.meta["groups"][c["name"]] = dict(n_variants=c["n_variants"], n_vcol=-(-max(c["n_variants"], 1) // vc),
                                              n_scol=-(-max(S, 1) // sc), n_chunks=len(c["offsets"]) - 1,
                                              compressed_bytes=c["offsets"][-1], raw_bytes=c["raw_bytes"])
        self._cur = None

    def close(self):
        json.dump(self.meta, open(os.path.join(self.path, "meta.json"), "w"), indent=1)


class GenotypeStore:
    """Reader.  Decoding runs on the GPU (hhgt_decompress_chunks); there is no CPU decode path in the
    product."""

    def __init__(self, path, ctx=None):
        self.path = path
        self.meta = json.load(open(os.path.join(path, "meta.json")))
        self.samples = self.meta["samples"]
        self._ctx = ctx
        self._idx = {s: i for i, s in enumerate(self.samples)}

    def _context(self):
        if self._ctx is None:
            from .device import Context
            self._ctx = Context(0)
        return self._ctx

    def groups(self):
        return list(self.meta["groups"])

    def variants(self, group):
        d = os.path.join(self.path, group)
        return (np.load(os.path.join(d, "start.npy")), np.load(os.path.join(d, "ref.npy")),
                np.load(os.path.join(d, "alt.npy")), json.load(open(os.path.join(d, "chrom_runs.json"))))

    def sample_row(self, group, sample):
        """int8 [n_variants, 2] for one sample: decodes the sample's chunk row on the GPU."""
        import torch
        g = self.meta["groups"][group]
        s = self._idx[sample] if isinstance(sample, str) else int(sample)
        sc, vc = self.meta["sc"], self.meta["vc"]
        scol, sin = divmod(s, sc)
        off = np.load(os.path.join(self.path, group, "offsets.npy"))
        mm = np.memmap(os.path.join(self.path, group, "chunks.bin"), dtype=np.uint8, mode="r")
        ids = [v * g["n_scol"] + scol for v in range(g["n_vcol"])]
        parts, rel = [], [0]
        for i in ids:
            parts.append(np.asarray(mm[int(off[i]):int(off[i + 1])]))
            rel.append(rel[-1] + parts[-1].size)
        ctx = self._context()
        src = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.uint8)).to(ctx.device)
        d_off = torch.tensor(rel, dtype=torch.int64, device=ctx.device)
        chunk_nbytes = sc * vc * 2
        out, bad = ctx.decompress(src, d_off, len(ids), chunk_nbytes, typesize=self.meta["typesize"],
                                  blocksize=self.meta["blocksize"])
        if bad:
            raise RuntimeError(f"{bad} corrupt chunk(s) in {group}")
        rows = out.view(torch.int8).view(len(ids), sc, vc, 2)[:, sin].reshape(-1, 2)
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
