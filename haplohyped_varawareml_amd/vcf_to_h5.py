"""`vcf_to_h5` — the reference's converter CLI and class on the MI355X path.

Same surface as /root/reference/src/haplohyped/vcf_to_h5.py: the six click options (:209-216), the
class VCFtoHDF5Converter(cohort_name, vcf_dir, out_dir, sample_list_path, cores, cxx_threads) with
.donor_ids / .chromosomes / .tmp_dir / read_sample_list / process_donor / merge_h5_files / run,
input naming DIR/chr{N}.filtered.vcf.gz (:151), contig chr{N} (:98), groups chr_{N} (:132).
Different inside: one device pass per chromosome FILE encodes every sample (the reference makes
len(donors) x 22 passes, :142-152,191-192), the genotypes are stored once as a cohort matrix of
Blosc-framed chunks (store.py: the working store; h5file.py: the same chunks as `OUT/{cohort}.h5`, an HDF5 file
with filter-32001 datasets written without h5py) instead of S x 22 compound datasets, and failures are not
swallowed (the reference drops worker exceptions, :191-192,204-205).
"""
import logging
import os
import shutil
import time
from typing import List

import click

logger = logging.getLogger("haplohyped.vcf_to_h5")


class VCFtoHDF5Converter:
    def __init__(self, cohort_name: str, vcf_dir: str, out_dir: str, sample_list_path: str, cores: int,
                 cxx_threads: int, donor_records=None):
        self.cohort_name = cohort_name
        self.vcf_dir = vcf_dir
        self.out_dir = out_dir
        self.sample_list_path = sample_list_path
        self.cores = cores                # host inflate threads per file (BGZF)
        self.cxx_threads = cxx_threads    # kept for CLI compatibility (the reference's value is unused too)
        # the reference's literal per-donor compound datasets in the .h5: None = only for cohorts of <= 32 donors;
        # HHGT_DONOR_RECORDS=yes|no overrides (the CLI keeps exactly the reference's six options)
        env = os.environ.get("HHGT_DONOR_RECORDS", "").lower()
        self.donor_records = donor_records if donor_records is not None else {"yes": True, "no": False}.get(env)
        self.donor_ids = self.read_sample_list(sample_list_path)
        self.chromosomes = range(1, 23)
        self.tmp_dir = os.path.join(out_dir, "tmp_files")
        os.makedirs(self.tmp_dir, exist_ok=True)
        self.stats = {}

    def read_sample_list(self, sample_list_path: str) -> List[str]:
        try:
            with open(sample_list_path, "r") as f:
                return [line.strip() for line in f]          # vcf_to_h5.py:70-71 (blank lines are kept)
        except FileNotFoundError as e:
            logger.error(f"Sample list file not found: {e}")
            raise

    @property
    def store_path(self):
        return os.path.join(self.out_dir, f"{self.cohort_name}.hhgt")

    @property
    def h5_path(self):
        """the reference's output file (vcf_to_h5.py:161)"""
        return os.path.join(self.out_dir, f"{self.cohort_name}.h5")

    def genotype_vcf_to_store(self, ctx, writer, data_path: str, chromosome: int):
        """one chromosome file -> group chr_{N} (all samples at once)"""
        from .device import BLOSC1      # filter 32001 = hdf5-blosc: Blosc-1 chunk framing
        from .pipeline import stream_file
        group = f"chr_{chromosome}"
        state = {"begun": False}

        def on_header(names):
            missing = [d for d in self.donor_ids if d and d not in names]
            if missing:   # cpp/vcfpp.h:373-377
                raise RuntimeError(f"Error parsing VCF file: the {len(missing)}-th sample are not in the VCF.\n"
                                   f"parameter samples:{missing[0]}")
            if writer.meta["samples"] != list(names):
                if writer.meta["groups"]:
                    raise RuntimeError(f"{data_path}: sample columns differ from the previous chromosome files")
                writer.meta["samples"] = list(names)
            writer.begin_group(group)
            state["begun"] = True

        def on_columns(G_cols, n_cols, framed):
            writer.add_chunks(framed[0], framed[1], G_cols.numel())

        fs = stream_file(ctx, data_path, region=f"chr{chromosome}", sc=writer.meta["sc"], vc=writer.meta["vc"],
                         n_threads=self.cores or 0, on_header=on_header, on_columns=on_columns,
                         on_variants=writer.add_variants, fmt=BLOSC1)
        writer.add_chrom_runs(fs.chrom_runs)
        writer.end_group()
        self.stats[group] = fs
        return fs

    def process_donor(self, donor_id: str) -> None:
        """kept for API compatibility: the cohort path has no per-donor pass"""
        logger.info(f"donor {donor_id}: encoded together with the whole cohort")

    def merge_h5_files(self) -> None:
        """nothing to merge: groups are written into one store as they finish (vcf_to_h5.py:154-180)"""

    def run(self):
        from .device import Context, DEFAULT_SC, DEFAULT_VC
        from .store import StoreWriter, export_h5
        t0 = time.time()
        ctx = Context(0)
        writer = StoreWriter(self.store_path, [], DEFAULT_SC, DEFAULT_VC, cohort_name=self.cohort_name,
                             donor_ids=[d for d in self.donor_ids if d], chunk_format="blosc1")
        try:
            for chromosome in self.chromosomes:
                vcf_file = os.path.join(self.vcf_dir, f"chr{chromosome}.filtered.vcf.gz")
                if not os.path.exists(vcf_file):
                    logger.warning(f"{vcf_file} not found; chromosome {chromosome} skipped")
                    continue
                fs = self.genotype_vcf_to_store(ctx, writer, vcf_file, chromosome)
                logger.info(f"chr{chromosome}: {fs.n_kept} SNPs x {fs.n_samples} samples in {fs.seconds:.2f}s "
                            f"({fs.n_lines / max(fs.seconds, 1e-9):.0f} lines/s), ratio "
                            f"{fs.raw_bytes / max(fs.compressed_bytes, 1):.2f}")
            writer.close()
            n_donors = len([d for d in self.donor_ids if d])
            per_donor = self.donor_records if self.donor_records is not None else n_donors <= 32
            export_h5(self.store_path, self.h5_path, donor_records=per_donor, ctx=ctx)   # OUT/{cohort}.h5 (h5py + hdf5plugin read it)
            logger.info(f"Total time taken: {time.time() - t0:.2f} seconds")
        finally:
            ctx.close()
            shutil.rmtree(self.tmp_dir, ignore_errors=True)
        return self.store_path


@click.command()
@click.option("--cohort_name", required=True, type=str, help="Cohort specific name")
@click.option("--vcf", required=True, type=str, help="Path to VCF files directory")
@click.option("--outdir", required=True, type=str, help="Path to results save folder")
@click.option("--sample_list", required=True, type=str, help="Path to sample list file")
@click.option("--cores", default=os.cpu_count(), type=int, help="Number of CPU cores to use")
@click.option("--cxx_threads", default=4, type=int, help="Number of threads to use in the C++ code")
def main(cohort_name, vcf, outdir, sample_list, cores, cxx_threads):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    VCFtoHDF5Converter(cohort_name=cohort_name, vcf_dir=vcf, out_dir=outdir, sample_list_path=sample_list,
                       cores=cores, cxx_threads=cxx_threads).run()


if __name__ == "__main__":
    main()
