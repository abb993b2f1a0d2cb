"""`RandomHaplotypeDataset` — the reference's consumer API on the MI355X path (BASELINE config 5).

Same constructor and return shapes as /root/reference/src/datasets/haplotype_dataset.py:30-114:
    RandomHaplotypeDataset(bed_file, hdf5_genotype_file, hdf5_reference_file, samples_file,
                           encode_spec=None, seed=42, batch_size=1, seq_length=1000)
    len(ds) == rows of the BED file;  ds[i] -> (hap1, hap2) float32 [batch_size, seq_length, C];  ds.close()
`hdf5_genotype_file` is the cohort store written by vcf_to_h5 (store.py); `hdf5_reference_file` is a FASTA
(or .npz of uint8 arrays keyed by contig).  Tensors are produced ON the GPU by hhgt_onehot_windows.

Semantics.  The reference module cannot run as shipped (broken imports,
`encode_sequence` undefined, variants overlaid on zeros with no window filter).  This build implements
what the code evidently intends and documents each choice:
  * window = [new_start, new_start + seq_length) with new_start from calculate_midpoint_region (:11-16);
  * bases start from the reference window (docs/ARCHITECTURE.md:140-145), variants with
    window_start <= start < window_end are overlaid: allele == 1 -> ALT, otherwise the VCF REF (:99-100);
  * channel of a base = its position in the encode spec's key order, upper-cased, non-ACGT -> 'N'
    (common_utils.py:84-103); default spec A,C,G,T,N (common_utils.py:73);
  * random draws follow the reference's call order (region, donor, chromosome; :59-61) from
    np.random.seed(seed); the chromosome is then taken from the BED row (the reference indexes the
    genotype file with an unrelated random chromosome, :65-71).
There are no reference outputs to pin this consumer to ("parity unpinned"); tests compare the GPU
tensors with a numpy restatement of the rules above.
"""
import ctypes as C
import json
import os
from collections import OrderedDict

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib
from .store import GenotypeStore


def calculate_midpoint_region(start, end, seq_length):
    """haplotype_dataset.py:11-16"""
    midpt = (start + end) // 2
    half_seq_length = seq_length // 2
    new_start = max(0, midpt - half_seq_length)
    new_end = midpt + half_seq_length
    return new_start, new_end


def parse_encode_dict(encode_spec):
    """base -> channel map with the behaviour of common_utils.py:62-79: nothing given = A,C,G,T,N in that order; a
    sequence (list, tuple or string) numbers its bases in order; a dict passes through; anything else is a TypeError
    with the reference's message"""
    if isinstance(encode_spec, dict) and encode_spec:
        return encode_spec
    if isinstance(encode_spec, (str, list, tuple)) or not encode_spec:
        return dict(zip(encode_spec or "ACGTN", range(len(encode_spec or "ACGTN"))))
    raise TypeError("Please input as dict, list or string!")


def channel_lut(encode_spec):
    """256-entry base byte -> channel (key order of the spec); 255 = no channel"""
    keys = list(parse_encode_dict(encode_spec).keys())
    lut = np.full(256, 255, np.uint8)
    n_chan = keys.index("N") if "N" in keys else 255
    lut[:] = n_chan                                   # anything that is not A,C,G,T counts as N
    for b in "ACGT":
        ch = keys.index(b) if b in keys else 255
        lut[ord(b)] = ch
        lut[ord(b.lower())] = ch                      # ignore_case=True (common_utils.py:88-97)
    return lut, len(keys)


def read_fasta(path):
    """-> {contig: uint8 array of bases}"""
    out, name, parts = {}, None, []
    with open(path, "rb") as f:
        data = f.read()
    for block in data.split(b">")[1:]:
        nl = block.find(b"\n")
        name = block[:nl].split()[0].decode()
        seq = np.frombuffer(block[nl + 1:], dtype=np.uint8)
        out[name] = seq[(seq != 10) & (seq != 13)].copy()
    return out


class ReferenceGenome:
    """haplotype_dataset.py:18-28 — `get_sequence(chrom, start, end)` returns |S1 bases"""

    def __init__(self, path, encode_spec=None, device=None, ctx=None):
        self.encode_spec = parse_encode_dict(encode_spec)
        self._store = None
        if os.path.isdir(str(path)) or str(path).endswith((".h5", ".hdf5")):
            # the one-hot reference written by fasta_encoder: its store directory, or OUT/reference_genome.h5
            from .fasta_encoder import ReferenceGenome as EncodedGenome
            self._store = EncodedGenome(hdf5_file=str(path), ctx=ctx)
            self.contigs = {c: None for c in self._store.contigs()}
        elif str(path).endswith(".npz"):
            z = np.load(path)
            self.contigs = {k: z[k] for k in z.files}
        else:
            self.contigs = read_fasta(path)
        self.device = device
        self._dev = {}

    def get_sequence(self, chrom, start, end):
        return self.host_bases(str(chrom))[start:end].view("|S1")

    def device_bases(self, chrom):
        if chrom not in self._dev:
            if self._store is not None:
                # one-hot rows -> base letters, on the device (columns are in the store's sorted order)
                m = self._store.contig_meta(chrom)
                rows = self._store.get_sequence_device(chrom, 0, m["length"])
                letters = torch.tensor([ord(c) for c in m["columns"]], dtype=torch.uint8, device=rows.device)
                self._dev[chrom] = letters[rows.argmax(dim=1)].contiguous()
            else:
                self._dev[chrom] = torch.from_numpy(self.contigs[chrom]).to(self.device)
        return self._dev[chrom]

    def host_bases(self, chrom):
        if self._store is not None:
            return self.device_bases(chrom).cpu().numpy()
        return self.contigs[chrom]

    def close(self):
        self._dev.clear()


class RandomHaplotypeDataset(Dataset):
    def __init__(self, bed_file, hdf5_genotype_file, hdf5_reference_file, samples_file, encode_spec=None, seed=42,
                 batch_size=1, seq_length=1000, ctx=None):
        from .device import Context
        self.bed = []
        with open(bed_file) as f:                       # :32 (tab separated, no header)
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) >= 3:
                    self.bed.append((p[0], int(p[1]), int(p[2])))
        self._own_ctx = ctx is None
        self.ctx = ctx or Context(0)
        self.store = GenotypeStore(hdf5_genotype_file, ctx=self.ctx)
        self.reference_genome = ReferenceGenome(hdf5_reference_file, encode_spec, device=self.ctx.device, ctx=self.ctx)
        self.encode_spec = parse_encode_dict(encode_spec)
        self.lut, self.n_channels = channel_lut(encode_spec)
        self.donor_ids = self.read_samples(samples_file)
        self.chromosomes = np.arange(1, 23)
        self.batch_size = batch_size
        self.seq_length = seq_length
        self.set_random_seed(seed)
        self.num_samples = len(self.bed)
        self._groups = {}
        self._rows = OrderedDict()
        self.last_items = []

    def read_samples(self, samples_file):
        with open(samples_file, "r") as f:
            return [line.strip() for line in f]

    def set_random_seed(self, seed):
        np.random.seed(seed)

    def __len__(self):
        return self.num_samples

    # ---- device-side caches -------------------------------------------------------------------
    def _group(self, group):
        if group not in self._groups:
            start, ref, alt, _ = self.store.variants(group)
            d = self.ctx.device
            self._groups[group] = dict(start=start, d_start=torch.from_numpy(start.view(np.int32)).to(d),
                                       d_ref=torch.from_numpy(ref).to(d), d_alt=torch.from_numpy(alt).to(d))
        return self._groups[group]

    def _donor_row(self, group, donor):
        key = (group, donor)
        if key not in self._rows:
            row = torch.from_numpy(self.store.sample_row(group, donor)).to(self.ctx.device).contiguous()
            self._rows[key] = row
            while len(self._rows) > 64:
                self._rows.popitem(last=False)
        self._rows.move_to_end(key)
        return self._rows[key]

    def __getitem__(self, idx):
        L, B = self.seq_length, self.batch_size
        items = (_lib.Window * B)()
        keep = []
        self.last_items = []
        for b in range(B):
            region_idx = np.random.randint(0, self.num_samples)          # :59
            donor_idx = np.random.randint(0, len(self.donor_ids))         # :60
            np.random.randint(0, len(self.chromosomes))                   # :61 (drawn, see module docstring)
            chrom, start, end = self.bed[region_idx]
            donor_id = self.donor_ids[donor_idx]
            new_start, _ = calculate_midpoint_region(start, end, L)       # :68
            digits = chrom[3:] if chrom.startswith("chr") else chrom
            group = f"chr_{digits}"
            w = items[b]
            ref = self.reference_genome.device_bases(chrom) if chrom in self.reference_genome.contigs else None
            w.ref_ptr = ref.data_ptr() if ref is not None else 0
            w.ref_len = ref.numel() if ref is not None else 0
            w.win_start = new_start
            lo = hi = 0
            if group in self.store.meta["groups"] and donor_id in self.store.samples:
                g = self._group(group)
                lo = int(np.searchsorted(g["start"], new_start, side="left"))
                hi = int(np.searchsorted(g["start"], new_start + L, side="left"))
                row = self._donor_row(group, donor_id)
                keep.append(row)
                w.var_start_ptr, w.var_ref_ptr, w.var_alt_ptr = g["d_start"].data_ptr(), g["d_ref"].data_ptr(), g["d_alt"].data_ptr()
                w.geno_ptr, w.geno_first = row.data_ptr(), 0
            w.var_lo, w.var_hi = lo, hi
            self.last_items.append(dict(chrom=chrom, group=group, donor=donor_id, start=new_start, var_lo=lo, var_hi=hi))
        d = self.ctx.device
        with torch.cuda.device(d):
            d_items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(d)
            hap1 = torch.empty((B, L, self.n_channels), dtype=torch.float32, device=d)
            hap2 = torch.empty((B, L, self.n_channels), dtype=torch.float32, device=d)
            _lib.check(self.ctx.lib.hhgt_onehot_windows(self.ctx.h, C.c_void_p(d_items.data_ptr()), B, L,
                                                        self.lut.ctypes.data, self.n_channels,
                                                        C.c_void_p(hap1.data_ptr()), C.c_void_p(hap2.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.current_stream().synchronize()     # d_items / keep stay alive until the kernels are done
        return hap1, hap2

    def close(self):
        self.reference_genome.close()
        self._rows.clear()
        self._groups.clear()
        if self._own_ctx:
            self.ctx.close()
