"""`vcf_to_h5` — the reference's converter CLI and class on the MI355X path.

Same surface as /root/reference/src/haplohyped/vcf_to_h5.py: the six click options (:209-216), the
class VCFtoHDF5Converter(cohort_name, vcf_dir, out_dir, sample_list_path, cores, cxx_threads) with
.donor_ids / .chromosomes / .tmp_dir / read_sample_list / process_donor / merge_h5_files / run,
input naming DIR/chr{N}.filtered.vcf.gz (:151), contig chr{N} (:98), groups chr_{N} (:132), output
OUT/{cohort}.h5 (:161).
Different inside: one device pass per chromosome FILE encodes every sample (the reference makes
len(donors) x 22 passes, :142-152,191-192), the genotypes are stored once as a cohort matrix of
Blosc-framed chunks (store.py: the working store; h5file.py: the same chunks as `OUT/{cohort}.h5`, an HDF5 file
with filter-32001 datasets written without h5py) instead of S x 22 compound datasets, and failures are not
swallowed (the reference drops worker exceptions, :191-192,204-205).

Multi-GPU: the reference fans independent (donor, chromosome) jobs over a thread pool
(:142-152,191-192).  Here the unit is the chromosome file: the files are assigned to the node's GPUs
longest-first (sharding.lpt_assign on file sizes), ONE PROCESS PER GPU encodes its files into a partial
store, and rank 0 merges the partial stores — a directory move per group, order-independent like
merge_h5_files (:154-180).  No genotype byte crosses between ranks: torch.distributed (gloo) carries only the
start/stop barriers and the per-group statistics.  `HHGT_GPUS=N` (default: every visible GPU) sets the
number of workers; N = 1 runs in-process.
"""
import json
import logging
import os
import shutil
import socket
import time
from typing import List

import click

logger = logging.getLogger("haplohyped.vcf_to_h5")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def merge_stores(parts, final_path, group_order):
    """partial stores (one per rank, disjoint groups) -> one store.  Groups are moved, not copied; the merged
    meta.json lists them in `group_order` (the single-process order), so the result does not depend on how the
    groups were dealt to the ranks."""
    metas = [json.load(open(os.path.join(p, "meta.json"))) for p in parts]
    owner = {g: i for i, m in enumerate(metas) for g in m["groups"]}
    if sum(len(m["groups"]) for m in metas) != len(owner):
        raise RuntimeError("merge_stores: a group was written by two ranks")
    with_groups = [m for m in metas if m["groups"]]
    base = dict(with_groups[0] if with_groups else metas[0])
    for m in with_groups[1:]:
        if m["samples"] != base["samples"]:
            raise RuntimeError("sample columns differ between chromosome files (ranks disagree on the header)")
    os.makedirs(final_path, exist_ok=True)
    groups = {}
    for g in group_order:
        if g not in owner:
            continue
        src = os.path.join(parts[owner[g]], g)
        dst = os.path.join(final_path, g)
        if os.path.exists(dst):
            shutil.rmtree(dst)
        shutil.move(src, dst)
        groups[g] = metas[owner[g]]["groups"][g]
    base["groups"] = groups
    json.dump(base, open(os.path.join(final_path, "meta.json"), "w"), indent=1)
    for p in parts:
        shutil.rmtree(p, ignore_errors=True)
    return final_path


def wait_workers(procs, poll=0.2):
    """joins the worker processes; as soon as one has exited non-zero the others are ended (they would otherwise sit in a
    collective until its timeout).  -> exit codes of the workers that failed"""
    live = list(procs)
    failed = False
    while live:
        for p in list(live):
            p.join(timeout=poll)
            if p.exitcode is not None:
                live.remove(p)
                failed = failed or p.exitcode != 0
        if failed:
            for p in live:          # the processes this function was given, by handle — never by name or pattern
                p.terminate()
            for p in live:
                p.join(timeout=30)
            live = []
    return [p.exitcode for p in procs if p.exitcode != 0]


def _default_stream():
    from .pipeline import stream_files
    return stream_files


def convert_rank(conv, rank, world, device, chromosomes, part_path, stream_fn=None, make_ctx=None, h5_path=None):
    """the work of one rank: its chromosome files -> groups of the (partial) store at part_path, all through ONE
    ingest engine (pipeline.stream_files): while chromosome k is being encoded, the host threads already inflate k+1.
    stream_fn / make_ctx exist so that the CPU test suite can drive the rank / merge logic without a GPU.
    h5_path: write OUT/{cohort}.h5 directly instead of a store (one rank, no store asked for: store.H5CohortWriter)."""
    from ._lib import BLOSC1      # filter 32001 = hdf5-blosc: Blosc-1 chunk framing
    from .device import DEFAULT_SC, DEFAULT_VC
    from .store import H5CohortWriter, StoreWriter
    stream_fn = stream_fn or _default_stream()
    real_device = make_ctx is None      # (decided before the default is filled in: a test's fake context has no GPU to ask)
    if make_ctx is None:
        from .device import Context
        make_ctx = Context
    # this rank's share of the host: its reader threads (the engine starts `n_threads` per open file) stay on the CPUs of
    # its GPU's NUMA node and the N ranks of a node do not oversubscribe it (N x --cores threads before)
    from .sharding import pin_rank
    devices = None
    if real_device:
        try:
            import torch
            n_dev = max(torch.cuda.device_count(), 1)
            devices = [r % n_dev for r in range(world)]      # worker_entry's rule: rank r drives GPU r % n_dev
        except Exception:
            devices = None
    host = pin_rank(rank, world, device if real_device else None, conv.cores, devices=devices)
    ctx = make_ctx(device)
    if h5_path:
        writer = H5CohortWriter(h5_path, [], DEFAULT_SC, DEFAULT_VC, cohort_name=conv.cohort_name,
                                donor_ids=[d for d in conv.donor_ids if d])
    else:
        writer = StoreWriter(part_path, [], DEFAULT_SC, DEFAULT_VC, cohort_name=conv.cohort_name,
                             donor_ids=[d for d in conv.donor_ids if d], chunk_format="blosc1")
    stats = {}
    jobs = [(os.path.join(conv.vcf_dir, f"chr{c}.filtered.vcf.gz"), f"chr{c}") for c in chromosomes]

    def on_header(i, names):
        missing = [d for d in conv.donor_ids if d and d not in names]
        if missing:   # cpp/vcfpp.h:373-377
            raise RuntimeError(f"Error parsing VCF file: the {len(missing)}-th sample are not in the VCF.\n"
                               f"parameter samples:{missing[0]}")
        if writer.meta["samples"] != list(names):
            if writer.meta["groups"]:
                raise RuntimeError(f"{jobs[i][0]}: sample columns differ from the previous chromosome files")
            writer.meta["samples"] = list(names)
        writer.begin_group(f"chr_{chromosomes[i]}")

    def on_end(i, fs):
        group = f"chr_{chromosomes[i]}"
        writer.add_chrom_runs(fs.chrom_runs)
        writer.end_group()
        conv.stats[group] = fs
        stats[group] = dict(n_kept=fs.n_kept, n_samples=fs.n_samples, n_lines=fs.n_lines, seconds=fs.seconds,
                            raw_bytes=fs.raw_bytes, compressed_bytes=fs.compressed_bytes, rank=rank, device=device,
                            n_threads=host["n_threads"], cpus=host["cpus"], numa_node=host["numa_node"])
        logger.info(f"[gpu {device}] chr{chromosomes[i]}: {fs.n_kept} SNPs x {fs.n_samples} samples in {fs.seconds:.2f}s "
                    f"({fs.n_lines / max(fs.seconds, 1e-9):.0f} lines/s), ratio "
                    f"{fs.raw_bytes / max(fs.compressed_bytes, 1):.2f}")

    try:
        if jobs:
            extra = {}
            if h5_path and real_device:      # the .h5 is written behind the engine: the chunk buffers are held until they are on file
                extra["hold_columns"] = True
            stream_fn(ctx, jobs, sc=writer.meta["sc"], vc=writer.meta["vc"], n_threads=host["n_threads"], fmt=BLOSC1,
                      on_header=on_header, on_variants=lambda i, a, b, c: writer.add_variants(a, b, c),
                      on_columns=(lambda i, g, n, framed, release=None: writer.add_chunks(framed[0], framed[1], g.numel(), **({"release": release} if release else {}))),
                      on_end=on_end, **extra)
        writer.close()
    finally:
        if hasattr(ctx, "close"):
            ctx.close()
    return stats


def worker_entry(rank, world, port, cfg, stream_fn=None, make_ctx=None):
    """process entry of rank `rank` (spawned by VCFtoHDF5Converter.run, or by the test suite): gloo process group
    for the barriers and the statistics, convert_rank for the work, rank 0 merges."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    conv = VCFtoHDF5Converter(cfg["cohort_name"], cfg["vcf_dir"], cfg["out_dir"], cfg["sample_list_path"], cfg["cores"],
                              cfg["cxx_threads"], n_gpus=world)
    import datetime
    # a rank that dies hard (segfault, GPU fault) never reaches the collectives: the others give up after 10 minutes
    # instead of gloo's default 30, and run() ends them as soon as it sees the dead one
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    try:
        n_dev = torch.cuda.device_count() if make_ctx is None else world
        device = rank % max(n_dev, 1)
        dist.barrier()
        t0 = time.time()
        err = None
        try:
            stats = convert_rank(conv, rank, world, device, cfg["plan"][rank], cfg["parts"][rank], stream_fn, make_ctx)
        except Exception as e:       # every rank must reach the collectives below
            stats, err = {}, f"rank {rank}: {type(e).__name__}: {e}"
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(stats=stats, err=err, seconds=time.time() - t0))
        errs = [g["err"] for g in gathered if g["err"]]
        if errs:
            raise RuntimeError("; ".join(errs))
        if rank == 0:
            merge_stores(cfg["parts"], cfg["store_path"], [f"chr_{c}" for c in cfg["chromosomes"]])
            allstats = {k: v for g in gathered for k, v in g["stats"].items()}
            json.dump(dict(world=world, plan=cfg["plan"], seconds=[g["seconds"] for g in gathered], groups=allstats),
                      open(os.path.join(cfg["store_path"], "ranks.json"), "w"), indent=1)
        dist.barrier()
    finally:
        dist.destroy_process_group()


class VCFtoHDF5Converter:
    def __init__(self, cohort_name: str, vcf_dir: str, out_dir: str, sample_list_path: str, cores: int,
                 cxx_threads: int, donor_records=None, n_gpus=None, keep_store=None):
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
        # one worker process per GPU (None: HHGT_GPUS, else every visible GPU); the working store is removed after
        # the export unless keep_store / HHGT_KEEP_STORE=1 (the .h5 alone serves every reader of this package)
        self.n_gpus = n_gpus if n_gpus is not None else (int(os.environ["HHGT_GPUS"]) if os.environ.get("HHGT_GPUS") else None)
        self.keep_store = keep_store if keep_store is not None else os.environ.get("HHGT_KEEP_STORE", "0") not in ("", "0")
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
        """one chromosome file -> group chr_{N} (all samples at once); the single-file form of what convert_rank
        does for a whole rank (the reference's per-(donor, chromosome) method, vcf_to_h5.py:79-140)"""
        from ._lib import BLOSC1      # filter 32001 = hdf5-blosc: Blosc-1 chunk framing
        from .pipeline import stream_file as stream_fn
        group = f"chr_{chromosome}"

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

        def on_columns(G_cols, n_cols, framed):
            writer.add_chunks(framed[0], framed[1], G_cols.numel())

        fs = stream_fn(ctx, data_path, region=f"chr{chromosome}", sc=writer.meta["sc"], vc=writer.meta["vc"],
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
        """nothing to merge per donor; the per-GPU partial stores are merged by merge_stores (vcf_to_h5.py:154-180)"""

    def present_chromosomes(self):
        out = []
        for chromosome in self.chromosomes:
            vcf_file = os.path.join(self.vcf_dir, f"chr{chromosome}.filtered.vcf.gz")
            if os.path.exists(vcf_file):
                out.append(chromosome)
            else:
                logger.warning(f"{vcf_file} not found; chromosome {chromosome} skipped")
        return out

    def plan(self, chromosomes, world):
        """chromosome files -> ranks, longest-processing-time-first on the file sizes"""
        from .sharding import lpt_assign
        sizes = [os.path.getsize(os.path.join(self.vcf_dir, f"chr{c}.filtered.vcf.gz")) for c in chromosomes]
        return [[chromosomes[i] for i in idx] for idx in lpt_assign(sizes, world)]

    def _config(self, chromosomes, world):
        return dict(cohort_name=self.cohort_name, vcf_dir=self.vcf_dir, out_dir=self.out_dir,
                    sample_list_path=self.sample_list_path, cores=self.cores, cxx_threads=self.cxx_threads,
                    chromosomes=list(chromosomes), plan=self.plan(chromosomes, world), store_path=self.store_path,
                    parts=[f"{self.store_path}.part{r}" for r in range(world)])

    def run(self):
        from .store import export_h5
        t0 = time.time()
        chromosomes = self.present_chromosomes()
        world = self.n_gpus
        if world is None:
            import torch
            world = max(torch.cuda.device_count(), 1)     # counting devices does not initialise the GPU
        world = max(1, min(int(world), max(len(chromosomes), 1)))
        if os.path.isdir(self.store_path):
            shutil.rmtree(self.store_path)
        n_donors = len([d for d in self.donor_ids if d])
        per_donor = self.donor_records if self.donor_records is not None else n_donors <= 32
        # one GPU, no working store asked for, no per-donor datasets (those are made from the store): the chunks go straight
        # into OUT/{cohort}.h5 as the engine hands them over — no store, no second copy of every chunk (HHGT_H5_DIRECT=0: the
        # store + export path, which N workers and the per-donor datasets use)
        direct = world == 1 and not self.keep_store and not per_donor and os.environ.get("HHGT_H5_DIRECT", "1") != "0"
        try:
            if world == 1:
                self.stats_by_group = convert_rank(self, 0, 1, 0, chromosomes, self.store_path, h5_path=self.h5_path if direct else None)
            else:
                # one process per GPU, started before this process touches a device
                import torch.multiprocessing as mp
                cfg = self._config(chromosomes, world)
                port = _free_port()
                ctx = mp.get_context("spawn")
                procs = [ctx.Process(target=worker_entry, args=(r, world, port, cfg)) for r in range(world)]
                for p in procs:
                    p.start()
                bad = wait_workers(procs)
                if bad:
                    for part in cfg["parts"]:
                        shutil.rmtree(part, ignore_errors=True)
                    raise RuntimeError(f"vcf_to_h5: {len(bad)} of {world} GPU workers failed (exit codes {bad})")
                self.stats_by_group = json.load(open(os.path.join(self.store_path, "ranks.json")))["groups"]
                logger.info(f"{world} GPU workers: plan {cfg['plan']}")
            if not per_donor:
                logger.warning(f"{n_donors} donors: the per-donor datasets donor_{{id}}/chr_{{N}}/snp_data of the reference "
                               f"layout are NOT written into {self.h5_path} (they would repeat the variant table per donor); "
                               f"the cohort matrix chr_{{N}}/genotype holds every genotype and VCFH5Reader serves the "
                               f"per-donor records from it.  Set HHGT_DONOR_RECORDS=yes to write them anyway.")
            if not direct:
                export_h5(self.store_path, self.h5_path, donor_records=per_donor)   # OUT/{cohort}.h5 (h5py + hdf5plugin read it)
                if not self.keep_store:
                    shutil.rmtree(self.store_path, ignore_errors=True)
            logger.info(f"Total time taken: {time.time() - t0:.2f} seconds; wrote {self.h5_path}")
        finally:
            shutil.rmtree(self.tmp_dir, ignore_errors=True)
        return self.h5_path


@click.command()
@click.option("--cohort_name", required=True, type=str, help="Cohort specific name")
@click.option("--vcf", required=True, type=str, help="Path to VCF files directory")
@click.option("--outdir", required=True, type=str, help="Path to results save folder")
@click.option("--sample_list", required=True, type=str, help="Path to sample list file")
@click.option("--cores", default=os.cpu_count(), type=int, help="Number of CPU cores to use")
@click.option("--cxx_threads", default=4, type=int, help="Number of threads to use in the C++ code")
def main(cohort_name, vcf, outdir, sample_list, cores, cxx_threads):
    """Writes OUTDIR/COHORT_NAME.h5.  One worker process per visible GPU (HHGT_GPUS=N overrides).  Cohorts of more
    than 32 donors get the cohort matrix chr_{N}/genotype only; HHGT_DONOR_RECORDS=yes adds the per-donor
    donor_{id}/chr_{N}/snp_data datasets of the reference layout.  HHGT_KEEP_STORE=1 keeps the working store."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    VCFtoHDF5Converter(cohort_name=cohort_name, vcf_dir=vcf, out_dir=outdir, sample_list_path=sample_list,
                       cores=cores, cxx_threads=cxx_threads).run()


if __name__ == "__main__":
    main()
