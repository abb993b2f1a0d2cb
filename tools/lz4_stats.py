"""Event counts of the LZ4 window loop on the bench workload (development tool).
Build the instrumented library first (here, no GPU needed):
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DHHGT_LZ4_STATS -Iinclude \
        -o build/libhhgt_stats.so haplohyped_varawareml_amd/csrc/*.hip -lz -lpthread
then on the GPU box:  HHGT_LIB=$PWD/build/libhhgt_stats.so python tools/lz4_stats.py [variants] [samples]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from haplohyped_varawareml_amd import _lib, device as dev, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
ctx = dev.Context(0)
L = _lib.load()
tab = synth.variant_table(1001, V, S)
text, nbytes = ctx.synth_fixed("chr1", tab, S, seed=1001)
res = ctx.encode_text(text, S, region="chr1", layout=dev.make_layout(S, V))
ctx.pad_tail(res)
out = (C.c_ulonglong * 8)()
L.hhgt_debug_lz4_stats(out, 1)
chunk = res.layout.sc * res.layout.vc * 2
c = ctx.compress(res.G, chunk, typesize=2, blocksize=dev.DEFAULT_BLOCKSIZE, fmt=dev.BLOSC2)
torch.cuda.synchronize()
L.hhgt_debug_lz4_stats(out, 0)
names = ["streams", "windows", "windows without candidate", "second batches", "extensions", "long extensions (>64 B)",
         "flushes (in loop)", "sequences"]
n = max(out[0], 1)
for k, v in zip(names, out):
    print("%-28s %12d   per stream %8.2f" % (k, v, v / n))
print("bytes per stream", res.G.numel() / n, " windows per stream", out[1] / n, " sequences per window", out[7] / max(out[1], 1))
