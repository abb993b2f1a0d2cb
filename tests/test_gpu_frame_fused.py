"""-m gpu: the one-launch framing (csrc/frame.hip, k_frame_fused: block sizes, chunk offsets by a decoupled look-back, copy;
HHGT_FRAME_FUSED=1 — measured slower in round 4, so not the default) writes byte for byte what the three launches
(k_frame_sizes / k_scan_u64 / k_frame_write) write: the same cases are framed by a child process that runs with the switch
on, and the SHA-256 of every output buffer and offset table is compared.  Cases: several hundred chunks (the look-back walks more than one 64-chunk step), incompressible chunks
between compressible ones (the memcpyed form changes the chunk's size), both header formats, typesize 35 and 1, a leftover
block, chunks of one block, and chunks framed from bit planes."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def cases():
    rng = np.random.default_rng(77)
    out = []

    def sparse(n, p=0.06):
        return (rng.random(n) < p).astype(np.uint8)

    # (name, data, chunk_nbytes, typesize, blocksize)
    d = sparse(700 * 16384)
    out.append(("700 chunks", d, 16384, 2, 8192))
    d = sparse(40 * 131072)
    d.reshape(40, 131072)[[3, 4, 17, 39]] = rng.integers(0, 256, (4, 131072), dtype=np.uint8)      # incompressible chunks
    out.append(("memcpyed between", d, 131072, 2, 8192))
    out.append(("typesize 35", sparse(35 * 4096 * 9, 0.2), 35 * 4096, 35, 35 * 512))
    out.append(("typesize 1, leftover", sparse(50 * 10000), 10000, 1, 4096))
    out.append(("one block per chunk", sparse(300 * 8192), 8192, 2, 8192))
    out.append(("typesize 4", sparse(64 * 65536, 0.1), 65536, 4, 2048))
    return out


def run_all():
    """-> {case name + format: sha256 of the framed bytes and of the offset table}"""
    import torch
    from haplohyped_varawareml_amd import device as dev, synth
    from tests.gpu_util import to_dev
    ctx = dev.Context(0)
    res = {}
    for name, data, chunk, ts, bs in cases():
        for fmt in (dev.BLOSC1, dev.BLOSC2):
            dst, off, total = ctx.compress(to_dev(data), chunk, typesize=ts, blocksize=bs, fmt=fmt)
            h = hashlib.sha256(dst[:total].cpu().numpy().tobytes())
            h.update(off.cpu().numpy().tobytes())
            res[f"{name} fmt{fmt}"] = h.hexdigest()
    # chunks framed from bit planes (the bench's and the engine's path), 2 x 7 chunks
    S, V = 400, 12000
    tab = synth.variant_table(9, V, S)
    text, _ = ctx.synth_fixed("chr9", tab, S, seed=9)
    lay = dev.make_layout(S, 16384, sc=64, vc=8192)
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
    r = dev.EncodeResult(None, lay, z(lay.v_capacity, torch.int32), None, z(lay.v_capacity, torch.uint8), z(lay.v_capacity, torch.uint8), 0, {}, [],
                         z(dev.planes_bytes(lay), torch.uint8))
    cur = z(1, torch.int64)
    ctx.encode_text_planes_async(text, S, r, cur, max_lines=V + 8, region="chr9").wait()
    ctx.pad_tail_planes_cursor(r, cur)
    dst, off, total = ctx.compress_planes(r, fmt=dev.BLOSC1)
    h = hashlib.sha256(dst[:total].cpu().numpy().tobytes())
    h.update(off.cpu().numpy().tobytes())
    res["planes"] = h.hexdigest()
    ctx.close()
    return res


def test_fused_framing_writes_the_same_bytes():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.environ.get("HHGT_FRAME_FUSED", "0") == "0", "this test compares the default (three launches) with a child that runs the fused kernel"
    mine = run_all()
    env = dict(os.environ, HHGT_FRAME_FUSED="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "tests.test_gpu_frame_fused"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    theirs = json.loads(out.stdout.strip().splitlines()[-1])
    assert theirs.pop("fused") == "1"
    assert set(mine) == set(theirs) and len(mine) >= 13
    for k in mine:
        assert mine[k] == theirs[k], f"{k}: the fused framing differs from the three-launch framing"


if __name__ == "__main__":
    r = run_all()
    r["fused"] = os.environ.get("HHGT_FRAME_FUSED", "0")
    print(json.dumps(r))
