"""helpers shared by the -m gpu parity tests: run the HIP path through the C ABI (via
haplohyped_varawareml_amd.device) and bring results back as numpy for comparison with the oracle."""
import numpy as np
import torch

from haplohyped_varawareml_amd import device as dev


def to_dev(buf):
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
    t = torch.empty(a.size + 16, dtype=torch.uint8, device="cuda")   # 16 B of slack keeps slicing aligned
    if a.size:
        t[:a.size] = torch.from_numpy(a.copy())
    return t[:a.size]


def gpu_encode(ctx, text, n_samples, region="", sc=dev.DEFAULT_SC, vc=dev.DEFAULT_VC, cap=None):
    """-> dict(G int8 [S, n_kept, 2], start, stop, ref, alt, stats, n_kept, res) like oracle.vcf_encode"""
    t = text if torch.is_tensor(text) else to_dev(text)
    lay = None
    if cap is not None or sc != dev.DEFAULT_SC or vc != dev.DEFAULT_VC:
        if cap is None:
            cap = t.numel() // (16 + 2 * max(n_samples, 0)) + 1
        lay = dev.make_layout(n_samples, cap, sc=sc, vc=vc)
    res = ctx.encode_text(t, n_samples, region=region, layout=lay)
    n = res.n_kept
    G = res.dense().cpu().numpy() if n_samples > 0 else np.zeros((0, n, 2), np.int8)
    return dict(G=np.ascontiguousarray(G), start=res.start[:n].cpu().numpy().view(np.uint32),
                stop=res.stop[:n].cpu().numpy().view(np.uint32), ref=res.ref[:n].cpu().numpy(),
                alt=res.alt[:n].cpu().numpy(), stats=res.stats, n_kept=n, res=res)


def assert_same_as_oracle(g, o):
    assert g["n_kept"] == o["n_kept"]
    assert np.array_equal(g["G"], o["G"]), "genotype matrix differs from the oracle"
    assert np.array_equal(g["start"], o["start"]) and np.array_equal(g["stop"], o["stop"])
    assert np.array_equal(g["ref"], o["ref"]) and np.array_equal(g["alt"], o["alt"])
    for k in ("n_records", "n_kept", "n_drop_region", "n_drop_filter", "n_haploid_padded"):
        assert g["stats"][k] == o["stats"][k], (k, g["stats"], o["stats"])


def split_chunks(dst, chunk_off, total):
    d = dst[:total].cpu().numpy()
    off = chunk_off.cpu().numpy()
    return [d[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
