"""development: time of the encode stage (planes or int8) on one chr1-sized shard, for the library HHGT_LIB names"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haplohyped_varawareml_amd import device as dev, synth

V, S = int(os.environ.get("ENC_V", 250000)), 2504
ctx = dev.Context(0)
tab = synth.variant_table(1001, V, S)
text, nbytes = ctx.synth_fixed("chr1", tab, S, seed=1001)
lay = dev.make_layout(S, V, vc=8192)
z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
for mode in sys.argv[1:] or ["planes"]:
    planes = mode == "planes"
    res = dev.EncodeResult(None if planes else z(dev.layout_bytes(lay), torch.uint8), lay, z(lay.v_capacity, torch.int32), None,
                           z(lay.v_capacity, torch.uint8), z(lay.v_capacity, torch.uint8), 0, {}, [],
                           z(dev.planes_bytes(lay), torch.uint8) if planes else None)
    cur = z(1, torch.int64)
    ctx.profile(True)
    for it in range(6):
        if it == 1:
            ctx.profile_reset()
        cur.zero_()
        (ctx.encode_text_planes_async if planes else ctx.encode_text_async)(text, S, res, cur, max_lines=V + 64, region="chr1")
    torch.cuda.synchronize()
    st = ctx.profile_read()
    ms = st["encode"]["ms"] / 5
    print(f"{os.path.basename(os.environ.get('HHGT_LIB', 'libhhgt.so')):28s} {mode:6s} encode {ms:.3f} ms  text {nbytes / ms / 1e9:.2f} TB/s read", flush=True)
