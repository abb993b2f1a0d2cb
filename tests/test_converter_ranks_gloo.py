"""CPU, world_size 2 over gloo: the multi-GPU mode of VCFtoHDF5Converter — chromosome files dealt to ranks
longest-first, one process per rank writing its own partial store, rank 0 merging — gives a store that is
byte-identical to the single-process one.  (No GPU here: the ranks use the oracle-backed stand-in of
tests/fake_pipeline.py for the device pass; tests/test_gpu_pipeline.py repeats this with the real one.)"""
import filecmp
import json
import os
import shutil
import socket

import numpy as np
import torch.multiprocessing as mp

from haplohyped_varawareml_amd import synth, vcf_to_h5
from haplohyped_varawareml_amd.reader import write_bgzf
from tests import fake_pipeline


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _same_tree(a, b):
    cmp = filecmp.dircmp(a, b)
    assert not cmp.left_only and not cmp.right_only, (cmp.left_only, cmp.right_only)
    for f in cmp.common_files:
        assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f
    for d in cmp.common_dirs:
        _same_tree(os.path.join(a, d), os.path.join(b, d))


def _make_inputs(tmp_path, golden_dir, fixture_golden):
    vcf_dir = tmp_path / "vcf"
    vcf_dir.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr22.filtered.vcf.gz"), vcf_dir / "chr22.filtered.vcf.gz")
    names = fixture_golden["samples"]
    write_bgzf(str(vcf_dir / "chr4.filtered.vcf.gz"), synth.render_mixed("chr4", 900, 3, seed=4, names=names))
    tab = synth.variant_table(7, 300, 3)
    text, _ = synth.render_fixed_numpy("chr7", tab, 3, seed=7, names=names)
    write_bgzf(str(vcf_dir / "chr7.filtered.vcf.gz"), text)
    return str(vcf_dir), os.path.join(golden_dir, "ipscs_samples_test.txt")


def test_two_ranks_equal_single_process(tmp_path, golden_dir, fixture_golden):
    vcf_dir, samples = _make_inputs(tmp_path, golden_dir, fixture_golden)
    # single process
    one = vcf_to_h5.VCFtoHDF5Converter("c", vcf_dir, str(tmp_path / "one"), samples, 2, 1, n_gpus=1)
    chroms = one.present_chromosomes()
    assert chroms == [4, 7, 22]
    vcf_to_h5.convert_rank(one, 0, 1, 0, chroms, one.store_path, fake_pipeline.stream_files, fake_pipeline.FakeCtx)
    # two ranks over gloo
    two = vcf_to_h5.VCFtoHDF5Converter("c", vcf_dir, str(tmp_path / "two"), samples, 2, 1, n_gpus=2)
    cfg = two._config(chroms, 2)
    assert sorted(c for r in cfg["plan"] for c in r) == chroms and all(cfg["plan"])     # a partition, nobody idle
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=vcf_to_h5.worker_entry, args=(r, 2, port, cfg, fake_pipeline.stream_files, fake_pipeline.FakeCtx))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ranks = json.load(open(os.path.join(two.store_path, "ranks.json")))
    assert ranks["world"] == 2 and sorted(ranks["groups"]) == ["chr_22", "chr_4", "chr_7"]
    assert {g["rank"] for g in ranks["groups"].values()} == {0, 1}
    # the host is dealt to the ranks: disjoint CPU sets, and each rank's reader-thread budget is its share of --cores
    # (2 // 2) — not the whole host per rank, which is what 8 ranks x all CPUs would have been
    by_rank = {g["rank"]: g for g in ranks["groups"].values()}
    assert not set(by_rank[0]["cpus"]) & set(by_rank[1]["cpus"])
    assert set(by_rank[0]["cpus"]) | set(by_rank[1]["cpus"]) <= set(os.sched_getaffinity(0))
    assert all(g["n_threads"] == 1 for g in by_rank.values())
    assert all(1 <= g["n_threads"] <= len(g["cpus"]) for g in by_rank.values())
    os.remove(os.path.join(two.store_path, "ranks.json"))
    assert not os.path.exists(cfg["parts"][0]) and not os.path.exists(cfg["parts"][1])
    _same_tree(one.store_path, two.store_path)
    meta = json.load(open(os.path.join(two.store_path, "meta.json")))
    assert list(meta["groups"]) == ["chr_4", "chr_7", "chr_22"]          # chromosome order, not rank order
    # and the merged store holds the golden matrix
    from oracle import oracle
    off = np.load(os.path.join(two.store_path, "chr_22", "offsets.npy"))
    raw = np.fromfile(os.path.join(two.store_path, "chr_22", "chunks.bin"), dtype=np.uint8)
    back = oracle.blosc_decompress(raw[off[0]:off[1]]).view(np.int8).reshape(64, 8192, 2)
    assert np.array_equal(back[:3, :1000], np.load(os.path.join(golden_dir, "fixture_G.npy")))


def test_failing_rank_fails_the_job(tmp_path, golden_dir, fixture_golden):
    """a rank that raises must not leave the others hanging in a collective, and the job must fail (the reference
    swallows worker exceptions, vcf_to_h5.py:191-192,204-205 — deliberately not mirrored)"""
    vcf_dir, samples = _make_inputs(tmp_path, golden_dir, fixture_golden)
    with open(os.path.join(vcf_dir, "chr7.filtered.vcf.gz"), "wb") as f:
        f.write(b"not a vcf\n")
    two = vcf_to_h5.VCFtoHDF5Converter("c", vcf_dir, str(tmp_path / "two"), samples, 2, 1, n_gpus=2)
    cfg = two._config(two.present_chromosomes(), 2)
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=vcf_to_h5.worker_entry, args=(r, 2, port, cfg, fake_pipeline.stream_files, fake_pipeline.FakeCtx))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode not in (0, None) for p in procs)
    assert not os.path.exists(os.path.join(two.store_path, "meta.json"))


def test_merge_rejects_mismatched_headers(tmp_path):
    from haplohyped_varawareml_amd.store import StoreWriter
    import pytest
    for i, names in enumerate((["a", "b"], ["a", "c"])):
        w = StoreWriter(str(tmp_path / f"p{i}"), names, 64, 8192, chunk_format="blosc1")
        w.begin_group(f"chr_{i + 1}")
        w.end_group()
        w.close()
    with pytest.raises(RuntimeError, match="sample columns differ"):
        vcf_to_h5.merge_stores([str(tmp_path / "p0"), str(tmp_path / "p1")], str(tmp_path / "m"), ["chr_1", "chr_2"])


def test_cpu_partition_rules():
    """sharding.partition_cpus: disjoint shares, NUMA-local where the topology is known, nobody without a CPU"""
    from haplohyped_varawareml_amd.sharding import _parse_cpulist, partition_cpus
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(64))
    # no topology: an even split
    shares = [partition_cpus(allowed, 8, r) for r in range(8)]
    assert all(len(s) == 8 for s in shares) and sorted(sum(shares, [])) == allowed
    # two sockets of 32, GPUs 0-3 on node 0 and 4-7 on node 1: every rank stays on its GPU's socket
    node_of = [0, 0, 0, 0, 1, 1, 1, 1]
    cpus_of = lambda n: list(range(32 * n, 32 * n + 32))
    shares = [partition_cpus(allowed, 8, r, node_of, cpus_of) for r in range(8)]
    assert all(len(s) == 8 for s in shares) and sorted(sum(shares, [])) == allowed
    assert all(set(shares[r]) <= set(cpus_of(node_of[r])) for r in range(8))
    # a restricted affinity mask (a container that grants CPUs 0-15 only): node 1 has no allowed CPU, its ranks share the rest
    shares = [partition_cpus(list(range(16)), 4, r, [0, 0, 1, 1], cpus_of) for r in range(4)]
    assert all(shares) and len({c for s in shares for c in s}) == sum(len(s) for s in shares)
    # one rank's node unknown
    shares = [partition_cpus(allowed, 4, r, [0, None, 1, 1], cpus_of) for r in range(4)]
    assert len({c for s in shares for c in s}) == sum(len(s) for s in shares) and all(shares)
    assert set(shares[0]) <= set(cpus_of(0)) and set(shares[2]) <= set(cpus_of(1))
    # fewer CPUs than ranks: shared
    assert partition_cpus([0, 1], 4, 3) == [0, 1]


def test_dead_worker_ends_the_others(tmp_path):
    """vcf_to_h5.wait_workers: one worker that dies hard (no collective reached) must not leave the job waiting for the
    gloo timeout — the survivors are ended as soon as the death is seen"""
    import time
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sleep_or_die, args=(r,)) for r in range(3)]
    t0 = time.time()
    for p in procs:
        p.start()
    bad = vcf_to_h5.wait_workers(procs)
    assert time.time() - t0 < 60 and len(bad) >= 1 and all(p.exitcode is not None for p in procs)


def _sleep_or_die(rank):
    import os
    import time
    if rank == 1:
        os._exit(7)
    time.sleep(600)
