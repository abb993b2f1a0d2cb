"""CPU, world_size 2 over gloo: the N>1 path of bench.py / the converter — shard plans partition the
22 chromosome shards, the job time is the max over ranks and the units are summed."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from haplohyped_varawareml_amd import sharding, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = synth.shard_sizes(3_000_000)
    mine, seed_off = sharding.plan(sizes, rank, world, "strong")
    wk, wseed = sharding.plan(sizes, rank, world, "weak")
    units = sum(sizes[i] for i in mine)
    dist.barrier()
    t, u = sharding.reduce_job(dist, 1.0 + rank, units)
    gathered = sharding.gather_objects(dist, mine)          # what bench.py's per_rank array is built with
    q.put((rank, mine, seed_off, wk, wseed, t, u, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_partition_and_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from haplohyped_varawareml_amd import synth
    sizes = synth.shard_sizes(3_000_000)
    all_shards = sorted(i for r in res for i in r[1])
    assert all_shards == list(range(22))                       # a partition: nothing lost, nothing twice
    loads = [sum(sizes[i] for i in r[1]) for r in res]
    assert max(loads) / (sum(loads) / world) < 1.02            # LPT balance of the GRCh38-proportional shards
    for r in res:
        assert r[5] == 2.0 and r[6] == 3_000_000.0             # max over ranks, sum over ranks
        assert r[3] == list(range(22)) and r[4] == 100_000 * r[0]   # weak: own cohort, own seeds
        assert r[7] == [res[0][1], res[1][1]]


def test_lpt_eight_gpus():
    from haplohyped_varawareml_amd import sharding, synth
    sizes = synth.shard_sizes(3_000_000)
    for n in (1, 2, 4, 8):
        a = sharding.lpt_assign(sizes, n)
        assert sorted(i for x in a for i in x) == list(range(22))
        loads = [sum(sizes[i] for i in x) for x in a]
        # <= 1.05 at the GPU counts the scaling bench runs (2: 1.0002, 4: 1.029, 8: 1.043): under the point where SURVEY 8e's
        # variant-range split of chr1-chr8 would be worth its second pass over the big files' headers
        assert max(loads) / (sum(loads) / n) <= 1.05
    for n in (3, 5, 6, 7):
        a = sharding.lpt_assign(sizes, n)
        loads = [sum(sizes[i] for i in x) for x in a]
        assert max(loads) / (sum(loads) / n) <= 1.10


def test_gather_objects_without_a_group():
    from haplohyped_varawareml_amd import sharding
    assert sharding.gather_objects(None, {"rank": 0}) == [{"rank": 0}]


def test_pin_rank_takes_the_rank_to_device_list(monkeypatch):
    """ranks sharing GPUs (rehearsals, more workers than devices): the NUMA node asked for is the node of the rank's DEVICE"""
    from haplohyped_varawareml_amd import sharding
    asked = []
    monkeypatch.setattr(sharding, "gpu_numa_node", lambda d: asked.append(d) or 0)
    monkeypatch.setattr(sharding, "node_cpus", lambda n: sorted(__import__("os").sched_getaffinity(0)))
    monkeypatch.setattr(__import__("os"), "sched_setaffinity", lambda pid, cpus: None)
    monkeypatch.setattr(sharding, "effective_cpus", lambda: 8)
    r = sharding.pin_rank(3, 4, device=1, devices=[0, 1, 0, 1])
    assert asked[0] == 1 and asked[1:] == [0, 1, 0, 1]
    assert r["n_threads"] >= 1

