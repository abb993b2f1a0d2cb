"""Does an HBM-bound kernel run beside k_lz4_blocks when both are queued on different streams?  (development probe)
A = torch elementwise pass over a big tensor (256-thread workgroups, no LDS); B = hhgt compress of one chr1-sized shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haplohyped_varawareml_amd import device as dev, synth

V, S = 230_000, 2504
ctx = dev.Context(0)
tab = synth.variant_table(1001, V, S)
text, nbytes = ctx.synth_fixed("chr1", tab, S, seed=1001)
res = ctx.encode_text(text, S, region="chr1", layout=dev.make_layout(S, V))
ctx.pad_tail(res)
chunk = res.layout.sc * res.layout.vc * 2
n_chunks = res.G.numel() // chunk
dst = torch.empty(n_chunks * (chunk + 32), dtype=torch.uint8, device="cuda")
off = torch.zeros(n_chunks + 1, dtype=torch.int64, device="cuda")
big = torch.zeros(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30, dtype=torch.float32, device="cuda")   # 4 GiB
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()


def A():
    with torch.cuda.stream(sa):
        big.add_(1.0)


def B():
    with torch.cuda.stream(sb):
        ctx.compress(res.G, chunk, typesize=2, blocksize=dev.DEFAULT_BLOCKSIZE, fmt=dev.BLOSC2, dst=dst, chunk_off=off, sync=False)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("A alone (HBM-bound, %.1f GB r+w): %.3f ms" % (big.numel() * 8 / 1e9, timed(A)))
print("B alone (lz4+frame of %d variants): %.3f ms" % (V, timed(B)))
print("A then B queued together: %.3f ms" % timed(lambda: (A(), B())))
print("B then A queued together: %.3f ms" % timed(lambda: (B(), A())))
