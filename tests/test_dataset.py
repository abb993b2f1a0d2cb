"""RandomHaplotypeDataset (BASELINE config 5).  CPU part restates the reference's helper tests
(/root/reference/tests/test_utils.py:11-32); the -m gpu part compares the device one-hot tensors with a
numpy restatement of the documented semantics (haplotype_dataset.py:11-16,86-110, common_utils.py:84-103).
No reference outputs exist for this consumer: parity unpinned (see dataset.py docstring)."""
import os

import numpy as np
import torch
import pytest


def test_parse_encode_dict_like_reference():
    from haplohyped_varawareml_amd.dataset import parse_encode_dict
    assert parse_encode_dict(None) == {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}          # test_utils.py:11-15
    assert parse_encode_dict(["A", "C", "G", "T"]) == {"A": 0, "C": 1, "G": 2, "T": 3}   # :17-21
    assert parse_encode_dict("ACGTN")["N"] == 4
    custom = {"A": 1, "C": 2, "G": 3, "T": 4}
    assert parse_encode_dict(custom) == custom                                          # :23-27
    with pytest.raises(TypeError):                                                      # :29-32
        parse_encode_dict(123)


def test_midpoint_region_and_lut():
    from haplohyped_varawareml_amd.dataset import calculate_midpoint_region, channel_lut
    assert calculate_midpoint_region(10_000_000, 10_001_000, 1000) == (10_000_000, 10_001_000)
    assert calculate_midpoint_region(100, 200, 1000) == (0, 650)
    lut, C = channel_lut(None)
    assert C == 5 and [lut[ord(c)] for c in "ACGTNacgtnRX-"] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 4, 4, 4]
    lut4, C4 = channel_lut("ACGT")
    assert C4 == 4 and lut4[ord("N")] == 255 and lut4[ord("g")] == 2


def independent_lut(spec):
    """channel of every base byte, restated here from the documented rule (common_utils.py:62-103: channels in
    the key order of the spec, default A,C,G,T,N; case-insensitive; anything else is N; a base without a channel
    gives an all-zero row) — deliberately not the product's channel_lut"""
    order = list(spec) if spec else ["A", "C", "G", "T", "N"]
    chan = {b: i for i, b in enumerate(order)}
    lut = np.full(256, chan.get("N", 255), np.uint8)
    for b in "ACGT":
        lut[ord(b)] = lut[ord(b.lower())] = chan.get(b, 255)
    return lut, len(order)


def numpy_expected(items, L, spec, vcf_text, sample_names, ref_bases):
    """documented semantics, one item at a time, plain numpy.  Every input is independent of the product: the
    variant table and the donor's genotype rows come from the ORACLE run on the source VCF text, the reference bases
    from the array the test wrote, the channel map from independent_lut; only the random (region, donor) picks are
    the dataset's (`items`)."""
    from oracle import oracle
    lut, C = independent_lut(spec)
    out1 = np.zeros((len(items), L, C), np.float32)
    out2 = np.zeros_like(out1)
    enc = {}
    for b, it in enumerate(items):
        seq = np.full(L, ord("N"), np.uint8)
        ref = ref_bases.get(it["chrom"])
        if ref is not None:
            a, e = it["start"], min(it["start"] + L, len(ref))
            if e > a:
                seq[: e - a] = ref[a:e]
        h = [seq.copy(), seq.copy()]
        if it["chrom"] not in enc:
            enc[it["chrom"]] = oracle.vcf_encode(vcf_text, len(sample_names), region=it["chrom"])
        o = enc[it["chrom"]]
        row = o["G"][sample_names.index(it["donor"])] if o["n_kept"] else np.zeros((0, 2), np.int8)
        for j in range(o["n_kept"]):
            off = int(o["start"][j]) - it["start"]
            if 0 <= off < L:
                for k in (0, 1):
                    h[k][off] = o["alt"][j] if row[j, k] == 1 else o["ref"][j]
        for k, dst in ((0, out1), (1, out2)):
            ch = lut[h[k]]
            ok = ch < C
            dst[b, np.nonzero(ok)[0], ch[ok]] = 1.0
    return out1, out2


@pytest.mark.gpu
@pytest.mark.parametrize("seq_length,batch,spec", [(1000, 4, None), (4096, 3, "ACGT"), (1001, 2, None)])
def test_dataset_matches_numpy_restatement(ctx, tmp_path, golden_dir, fixture_text, fixture_golden, seq_length, batch, spec):
    import shutil
    from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
    from haplohyped_varawareml_amd.dataset import RandomHaplotypeDataset
    vcf_dir = tmp_path / "vcf"
    vcf_dir.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr22.filtered.vcf.gz"), vcf_dir / "chr22.filtered.vcf.gz")
    samples = os.path.join(golden_dir, "ipscs_samples_test.txt")
    store = VCFtoHDF5Converter("c", str(vcf_dir), str(tmp_path / "out"), samples, 2, 1).run()
    # synthetic reference covering the fixture's coordinates (the reference's chr22.fasta is 1 Mbp while
    # its VCF/BED sit at 10-20 Mbp): 20 Mbp of random bases with a few n/N
    rng = np.random.default_rng(1)
    seq = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.integers(0, 9, 20_100_000)]
    np.savez(tmp_path / "ref.npz", chr22=seq)
    ds = RandomHaplotypeDataset(os.path.join(golden_dir, "test_regions.bed"), store, str(tmp_path / "ref.npz"), samples,
                                encode_spec=spec, seed=42, batch_size=batch, seq_length=seq_length, ctx=ctx)
    assert len(ds) == 20                                   # BED rows (test_integration.py:46-55)
    C = 5 if spec is None else len(spec)
    n_var_total = 0
    for _ in range(3):
        h1, h2 = ds[0]
        assert h1.shape == h2.shape == (batch, seq_length, C) and h1.dtype == h2.dtype == __import__("torch").float32
        e1, e2 = numpy_expected(ds.last_items, seq_length, spec, fixture_text, fixture_golden["samples"], {"chr22": seq})
        assert np.array_equal(h1.cpu().numpy(), e1) and np.array_equal(h2.cpu().numpy(), e2)
        assert float(h1.sum()) <= batch * seq_length
        n_var_total += sum(it["var_hi"] - it["var_lo"] for it in ds.last_items)
    assert n_var_total > 0 or seq_length < 2000
    ds.close()


@pytest.mark.gpu
def test_dataset_config5_shape(ctx, tmp_path, golden_dir, fixture_text, fixture_golden):
    """BASELINE config 5: seq_length=131072, batch=32 -> two [32, 131072, 5] float32 tensors on the GPU"""
    import shutil
    import torch
    from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
    from haplohyped_varawareml_amd.dataset import RandomHaplotypeDataset
    vcf_dir = tmp_path / "vcf"
    vcf_dir.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr22.filtered.vcf.gz"), vcf_dir / "chr22.filtered.vcf.gz")
    samples = os.path.join(golden_dir, "ipscs_samples_test.txt")
    store = VCFtoHDF5Converter("c", str(vcf_dir), str(tmp_path / "out"), samples, 2, 1).run()
    rng = np.random.default_rng(2)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 20_200_000)]
    np.savez(tmp_path / "ref.npz", chr22=seq)
    ds = RandomHaplotypeDataset(os.path.join(golden_dir, "test_regions.bed"), store, str(tmp_path / "ref.npz"), samples,
                                seed=42, batch_size=32, seq_length=131072, ctx=ctx)
    h1, h2 = ds[0]
    assert h1.is_cuda and h1.shape == (32, 131072, 5)
    # every position has exactly one hot channel (reference bases are ACGT here)
    assert torch.equal(h1.sum(-1), torch.ones(32, 131072, device=h1.device))
    assert torch.equal(h2.sum(-1), torch.ones(32, 131072, device=h1.device))
    e1, e2 = numpy_expected(ds.last_items[:2], 131072, None, fixture_text, fixture_golden["samples"], {"chr22": seq})
    assert np.array_equal(h1[:2].cpu().numpy(), e1) and np.array_equal(h2[:2].cpu().numpy(), e2)
    assert sum(it["var_hi"] - it["var_lo"] for it in ds.last_items) > 32     # windows do contain variants
    ds.close()


@pytest.mark.gpu
def test_dataset_reads_fasta_encoder_store(ctx, tmp_path, golden_dir, fixture_text, fixture_golden):
    """full chain of the reference's README: vcf_to_h5 + fasta_encoder -> RandomHaplotypeDataset"""
    import shutil
    from click.testing import CliRunner
    from haplohyped_varawareml_amd import fasta_encoder
    from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
    from haplohyped_varawareml_amd.dataset import RandomHaplotypeDataset
    vcf_dir = tmp_path / "vcf"
    vcf_dir.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr22.filtered.vcf.gz"), vcf_dir / "chr22.filtered.vcf.gz")
    samples = os.path.join(golden_dir, "ipscs_samples_test.txt")
    store = VCFtoHDF5Converter("c", str(vcf_dir), str(tmp_path / "out"), samples, 2, 1).run()
    rng = np.random.default_rng(7)
    seq = b"".join(rng.choice([b"A", b"C", b"G", b"T", b"N", b"g"], 20_050_000).tolist())
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as f:
        f.write(b">chr22\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + b"\n")
    res = CliRunner().invoke(fasta_encoder.main, ["--fasta", str(fa), "--outdir", str(tmp_path / "out"), "--cores", "2"])
    assert res.exit_code == 0, res.output
    ref_store = str(tmp_path / "out" / "reference_genome.hhgt")
    ds = RandomHaplotypeDataset(os.path.join(golden_dir, "test_regions.bed"), store, ref_store, samples, seed=1,
                                batch_size=4, seq_length=2000, ctx=ctx)
    h1, h2 = ds[0]
    # expected values from the FASTA bytes the test wrote (the one-hot rule is case-insensitive), not from the store
    up = np.frombuffer(seq.upper(), dtype=np.uint8)
    e1, e2 = numpy_expected(ds.last_items, 2000, None, fixture_text, fixture_golden["samples"],
                            {"chr22": np.frombuffer(seq, dtype=np.uint8)})
    assert np.array_equal(h1.cpu().numpy(), e1) and np.array_equal(h2.cpu().numpy(), e2)
    assert np.array_equal(ds.reference_genome.host_bases("chr22"), up)
    ds.close()
    # the same dataset over the exported HDF5 file (the reference's `hdf5_genotype_file` argument) instead of the store
    ds5 = RandomHaplotypeDataset(os.path.join(golden_dir, "test_regions.bed"), str(tmp_path / "out" / "c.h5"),
                                 str(tmp_path / "out" / "reference_genome.h5"), samples,
                                 seed=1, batch_size=4, seq_length=2000, ctx=ctx)
    g1, g2 = ds5[0]
    assert torch.equal(g1, h1) and torch.equal(g2, h2)
    ds5.close()
    # the reference's artefact name: OUT/reference_genome.h5 with /{chrom}/sequence behind filter 32001
    h5 = str(tmp_path / "out" / "reference_genome.h5")
    assert os.path.exists(h5)
    from tests.test_h5file import h5check, have_h5py
    if have_h5py():
        import json
        from oracle import oracle
        got = h5check(h5, tmp_path, "chr22/sequence")
        meta = json.loads(str(got["chr22/sequence|meta"]))
        assert meta["shape"] == [len(seq), 5] and meta["dtype"] == "int8" and meta["filters"][0][0] == 32001
        assert b"".join(got["chr22/columns"].tolist()) == b"ACGNT"
        rows = meta["chunks"][0]
        first = oracle.blosc_decompress(got["chr22/sequence|chunk|0,0"]).reshape(rows, 5)
        want = np.zeros((rows, 5), np.uint8)
        want[np.arange(rows), np.array([b"ACGNT".index(c) for c in up[:rows].tobytes()])] = 1
        assert np.array_equal(first, want)
