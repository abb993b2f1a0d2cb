"""Drop-in for the reference's pybind11 module `parse_vcf` (/root/reference/cpp/parse_vcf.cpp:116-124).

    VCFLoader().load_vcf(in_vcf, sample, chrom="")            -> list[(chrom, start, stop, ref, alt, phase1, phase2)]
    VCFLoader().load_vcf_without_sample(in_vcf, chrom="")     -> list[(chrom, start, stop, ref, alt)]
plus the module-level spellings `parse_vcf.load_vcf(...)` that the reference's own caller uses
(/root/reference/src/haplohyped/vcf_to_h5.py:101) although its binding never defined them.

The reference re-opens and re-tokenises the file for every (sample, chromosome); here ONE device pass
per (file, chromosome) encodes every sample (G[s, v, 2]) and is kept on the GPU, so the S calls of the
reference's donor loop cost one pass.  Errors surface as RuntimeError, as in parse_vcf.cpp:63-66.
Differences by design: a missing tabix index is not an error (the reference asserts, vcfpp.h:1443);
haploid calls yield -9 for the second allele instead of tripping an assert (parse_vcf.cpp:46).
"""
import os
import threading
from collections import OrderedDict

import numpy as np

_CACHE = OrderedDict()
_CACHE_MAX = 2
_ctx = None
# The reference calls load_vcf from ThreadPoolExecutor workers (vcf_to_h5.py:191-192); its pybind11 binding holds
# the GIL for the whole call, so those calls never overlap.  ctypes and torch release the GIL, and an hhgt context is
# one-thread-at-a-time (include/hhgt.h), so the same serialisation is made explicit here: one lock around the cache
# lookup, the device pass and the insert.  A second thread asking for the same (file, chromosome) waits and then
# hits the cache instead of encoding the file again.
_LOCK = threading.RLock()


def _context():
    global _ctx
    with _LOCK:
        if _ctx is None:
            from .device import Context
            _ctx = Context(0)
        return _ctx


def _encoded(in_vcf, chrom, sites_only):
    with _LOCK:
        return _encoded_locked(in_vcf, chrom, sites_only)


def _encoded_locked(in_vcf, chrom, sites_only):
    from .pipeline import encode_file_resident
    st = os.stat(in_vcf)
    key = (os.path.abspath(in_vcf), st.st_mtime_ns, st.st_size, chrom or "", bool(sites_only))
    if key in _CACHE:
        _CACHE.move_to_end(key)
        return _CACHE[key]
    G, start, ref, alt, fs = encode_file_resident(_context(), in_vcf, region=chrom or "", sites_only=sites_only)
    runs = fs.chrom_runs
    bounds = [r[0] for r in runs] + [len(start)]
    chroms = np.empty(len(start), dtype=object)
    for (a, name), b in zip(runs, bounds[1:]):
        chroms[a:b] = name
    entry = dict(G=G, start=start, ref=ref, alt=alt, samples=fs.samples, chroms=chroms)
    _CACHE[key] = entry
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return entry


class VCFLoader:
    """Stateless, like the reference's class (cpp/parse_vcf.cpp:19-21)."""

    def load_vcf(self, in_vcf, sample, chrom=""):
        try:
            if not os.path.exists(in_vcf):
                raise RuntimeError(f"cannot open {in_vcf}")
            e = _encoded(in_vcf, chrom, False)
            if sample not in e["samples"]:
                # cpp/vcfpp.h:373-377
                raise RuntimeError("the 1-th sample are not in the VCF.\nparameter samples:" + str(sample))
            s = e["samples"].index(sample)
            with _LOCK:
                ph = e["G"][s].cpu().numpy()
        except RuntimeError as ex:
            msg = str(ex)
            if not msg.startswith("Error parsing VCF file"):
                msg = "Error parsing VCF file: " + msg
            raise RuntimeError(msg) from None
        start = e["start"]
        out = list(zip(e["chroms"].tolist(), start.tolist(), (start + 1).tolist(),
                       [chr(c) for c in e["ref"]], [chr(c) for c in e["alt"]],
                       ph[:, 0].tolist(), ph[:, 1].tolist()))
        print(f"Loaded {len(out)} SNPs for sample {sample} and chromosome {chrom}")   # parse_vcf.cpp:69
        return out

    def load_vcf_without_sample(self, in_vcf, chrom=""):
        try:
            if not os.path.exists(in_vcf):
                raise RuntimeError(f"cannot open {in_vcf}")
            e = _encoded(in_vcf, chrom, True)
        except RuntimeError as ex:
            msg = str(ex)
            if not msg.startswith("Error parsing VCF file"):
                msg = "Error parsing VCF file: " + msg
            raise RuntimeError(msg) from None
        start = e["start"]
        out = list(zip(e["chroms"].tolist(), start.tolist(), (start + 1).tolist(),
                       [chr(c) for c in e["ref"]], [chr(c) for c in e["alt"]]))
        print(f"Loaded {len(out)} SNPs for chromosome {chrom}")                        # parse_vcf.cpp:111
        return out


def load_vcf(in_vcf, sample, chrom=""):
    return VCFLoader().load_vcf(in_vcf, sample, chrom)


def load_vcf_without_sample(in_vcf, chrom=""):
    return VCFLoader().load_vcf_without_sample(in_vcf, chrom)
