"""Per-chromosome sharding across the GPUs of one node: shards are independent
(the reference already treats (donor, chromosome) pairs as independent jobs,
/root/reference/src/haplohyped/vcf_to_h5.py:142-152,191-192), so there is NO data-path collective;
torch.distributed is used only for the start/stop barrier and for reducing the timing."""
import os


def lpt_assign(sizes, n_ranks):
    """longest-processing-time-first bin packing: -> list (per rank) of sorted shard indices"""
    loads = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = loads.index(min(loads))
        out[r].append(i)
        loads[r] += sizes[i]
    return [sorted(o) for o in out]


def plan(sizes, rank, world, scaling="weak"):
    """-> (shard indices for this rank, seed offset).  weak: every rank owns a full cohort of its own
    (per-GPU work fixed); strong: the shards of ONE cohort are split over the ranks."""
    if scaling == "strong":
        return lpt_assign(sizes, world)[rank], 0
    return list(range(len(sizes))), 100_000 * rank


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def reduce_job(dist, seconds, units, device=None):
    """max over ranks of the elapsed time, sum over ranks of the units processed"""
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_objects(dist, obj):
    """every rank's `obj` on every rank, in rank order ([obj] without a process group): what makes an N-GPU line say which
    rank was the slow one (the reductions above only keep max and sum)"""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


# ---------------------------------------------------------------------------------------------------------------
# host cores of one node, dealt to the ranks ("NUMA-pin the inflate threads to the GPU's socket").
# N ranks that each start `all CPUs` reader threads oversubscribe the host N times — and the host inflate is exactly
# where per-chromosome scaling breaks first.  Every rank pins itself (and thereby every thread it starts: the reader's
# default thread count is the size of the affinity mask) to its share of the CPUs of its GPU's NUMA node.
# ---------------------------------------------------------------------------------------------------------------
def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index):
    """NUMA node of GPU `device_index` (/sys/bus/pci/devices/<bdf>/numa_node), or None when the platform does not say"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        return node if node >= 0 else None
    except Exception:
        return None


def node_cpus(node):
    try:
        return _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception:
        return None


def partition_cpus(allowed, world, rank, nodes=None, cpus_of_node=None):
    """CPUs of `allowed` that rank `rank` of `world` may use.  nodes[r] = NUMA node of rank r's GPU (None: unknown),
    cpus_of_node(node) -> CPUs of that node.  Ranks whose GPUs sit on the same node share that node's allowed CPUs
    evenly; a rank whose node is unknown (or has no allowed CPU) takes its even share of what the others leave.
    The sets of different ranks are disjoint and nobody is left without a CPU while there are at least `world`."""
    allowed = sorted(set(allowed))
    world = max(int(world), 1)
    if len(allowed) < world:                       # fewer CPUs than ranks: everybody shares everything
        return allowed
    shares = {}
    taken = set()
    if nodes is not None and cpus_of_node is not None:
        by_node = {}
        for r, n in enumerate(nodes):
            if n is not None:
                by_node.setdefault(n, []).append(r)
        for n, ranks in sorted(by_node.items()):
            cpus = [c for c in (cpus_of_node(n) or []) if c in set(allowed) and c not in taken]
            if len(cpus) < len(ranks):
                continue                            # not enough CPUs on that node: these ranks fall through to the even split
            per = min(len(cpus) // len(ranks), len(allowed) // world)   # never more than the fair share: the others need CPUs too
            for i, r in enumerate(ranks):
                shares[r] = cpus[i * per:(i + 1) * per]
                taken.update(shares[r])
    rest_ranks = [r for r in range(world) if r not in shares]
    rest = [c for c in allowed if c not in taken]
    if rest_ranks:
        per = max(len(rest) // len(rest_ranks), 1)
        for i, r in enumerate(rest_ranks):
            shares[r] = rest[i * per:(i + 1) * per] or rest[-1:]
    return shares[rank]


def effective_cpus():
    """CPUs' worth of time the process is granted: the affinity mask capped by the cgroup quota (libhhgt's
    hhgt_effective_cpus: a GPU box may show 256 hardware threads and grant 16)"""
    try:
        import ctypes
        from . import _lib
        L = _lib.load()
        L.hhgt_effective_cpus.restype = ctypes.c_int
        return max(int(L.hhgt_effective_cpus()), 1)
    except Exception:
        return max(len(os.sched_getaffinity(0)), 1)


def pin_rank(rank, world, device=None, cores=None, devices=None):
    """Pins the calling process to its share of the host CPUs (see partition_cpus) and returns
    dict(cpus, n_threads, numa_node, granted).  The thread budget of a rank is its share of what the host GRANTS
    (effective_cpus() // world), never more than its pinned CPUs, and never more than cores // world when the caller
    names a total (the reference's --cores).  World 1: the affinity is left alone.
    devices[r] = GPU index of rank r (default: rank r drives GPU r; ranks sharing GPUs — rehearsals, more workers than
    devices — name theirs, e.g. [r % n_dev for r in range(world)])."""
    allowed = sorted(os.sched_getaffinity(0))
    granted = min(effective_cpus(), len(allowed))
    node = gpu_numa_node(device) if device is not None else None
    if world <= 1:
        n = granted if not cores else max(1, min(int(cores), granted))
        return dict(cpus=allowed, n_threads=n, numa_node=node, granted=granted)
    if devices is None:
        devices = list(range(world))
    nodes = [gpu_numa_node(devices[r]) if device is not None and r < len(devices) else None for r in range(world)]
    cpus = partition_cpus(allowed, world, rank, nodes, node_cpus)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    n = max(1, min(len(cpus), granted // world))
    if cores:
        n = max(1, min(n, int(cores) // world))
    return dict(cpus=cpus, n_threads=n, numa_node=node, granted=granted)
