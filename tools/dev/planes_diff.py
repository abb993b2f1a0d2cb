"""development: where do chunks compressed from planes differ from chunks compressed from the expanded bytes?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from haplohyped_varawareml_amd import device as dev, synth
from tests.gpu_util import split_chunks, to_dev
from tests.test_gpu_planes import _encode_planes

ctx = dev.Context(0)
S, V = 400, 12000
text = ctx.synth_mixed("chr2", synth.mixed_table(2, V, S), S, seed=2)[0].cpu().numpy()
lay = dev.make_layout(S, 12000, sc=64, vc=8192)
res, n, _ = _encode_planes(ctx, text, S, lay, "chr2", 2, with_g=False, max_lines=lambda t: t.numel() // 16 + 8)
cn = 64 * 8192 * 2
raw = ctx.planes_expand(res)
def run_p():
    d, o, t = ctx.compress_planes(res, fmt=dev.BLOSC1)
    return split_chunks(d, o, t)
def run_b():
    d, o, t = ctx.compress(raw, cn, typesize=2, blocksize=8192, fmt=dev.BLOSC1)
    return split_chunks(d, o, t)
p1, p2, b1, b2 = run_p(), run_p(), run_b(), run_b()
same = lambda x, y: all(np.array_equal(a, b) for a, b in zip(x, y))
print("planes twice equal:", same(p1, p2), " bytes twice equal:", same(b1, b2), " planes==bytes:", same(p1, b1))
rawh = raw.cpu().numpy()
for ci, (a, b) in enumerate(zip(p1, b1)):
    if np.array_equal(a, b):
        continue
    nb = cn // 8192
    bsa = a[16:16 + 4 * nb].view(np.int32); bsb = b[16:16 + 4 * nb].view(np.int32)
    for k in range(nb):
        qa, qb = int(bsa[k]), int(bsb[k])
        for st in range(2):
            ca = int(a[qa:qa + 4].view(np.int32)[0]); cb = int(b[qb:qb + 4].view(np.int32)[0])
            sa = a[qa + 4:qa + 4 + ca]; sb = b[qb + 4:qb + 4 + cb]
            if ca != cb or not np.array_equal(sa, sb):
                plane = rawh[ci * cn + k * 8192:ci * cn + (k + 1) * 8192][st::2]
                print(f"chunk {ci} block {k} stream {st}: csize planes {ca} bytes {cb}; ones {(plane==1).sum()} missing {(plane==0xF7).sum()} other {((plane>1)&(plane!=0xF7)).sum()}")
            qa += 4 + ca; qb += 4 + cb
