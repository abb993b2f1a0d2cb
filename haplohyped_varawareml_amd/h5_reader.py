"""`VCFH5Reader` — the reference's reader class (/root/reference/src/utils/h5_reader.py:5-46) over the
cohort store.  fetch_genotypes(donor_id, chromosome) returns the reference's per-donor compound records
(dtype of vcf_to_h5.py:119-127) whichever dataset name the caller meant: the reference's writer says
`snp_data` (vcf_to_h5.py:134), its reader says `genotype` (h5_reader.py:40) — both resolve here."""
from .store import GenotypeStore


class VCFH5Reader:
    def __init__(self, h5_file, ctx=None):
        self.h5_file = h5_file
        self.store = GenotypeStore(h5_file, ctx=ctx)

    def fetch_genotypes(self, donor_id, chromosome):
        group = f"chr_{chromosome}"
        if group not in self.store.meta["groups"] or donor_id not in self.store.samples:
            raise KeyError(f"No data found for donor_{donor_id}/chr_{chromosome}")     # h5_reader.py:42-43
        return self.store.snp_records(group, donor_id)

    def close(self):
        pass
