#!/usr/bin/env python3
"""BASELINE config 5: RandomHaplotypeDataset one-hot decode, seq_length=131072, batch=32 -> windows/s and
the HBM write rate of hhgt_onehot_windows (not the driver's metric)."""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from haplohyped_varawareml_amd import device as dev
from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
from haplohyped_varawareml_amd.dataset import RandomHaplotypeDataset

g = os.path.join(ROOT, "tests", "golden")
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "vcf"))
shutil.copy(os.path.join(g, "chr22.filtered.vcf.gz"), os.path.join(tmp, "vcf", "chr22.filtered.vcf.gz"))
samples = os.path.join(g, "ipscs_samples_test.txt")
store = VCFtoHDF5Converter("c", os.path.join(tmp, "vcf"), os.path.join(tmp, "out"), samples, 2, 1).run()
rng = np.random.default_rng(2)
np.savez(os.path.join(tmp, "ref.npz"), chr22=np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 20_200_000)])
ctx = dev.Context(0)
ds = RandomHaplotypeDataset(os.path.join(g, "test_regions.bed"), store, os.path.join(tmp, "ref.npz"), samples,
                            seed=42, batch_size=32, seq_length=131072, ctx=ctx)
for _ in range(3):
    ds[0]
ctx.profile(True); ctx.profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N):
    h1, h2 = ds[0]
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
st = ctx.profile_read()
out_bytes = 2 * h1.numel() * 4
print(json.dumps(dict(batch=32, seq_length=131072, channels=5, ms_per_batch=dt * 1e3, windows_per_s=32 / dt,
                      onehot_kernel_ms=st["onehot"]["ms"] / N, onehot_write_GBps=out_bytes / (st["onehot"]["ms"] / N * 1e-3) / 1e9)))
shutil.rmtree(tmp)
