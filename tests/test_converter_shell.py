"""CPU: the converter shell behaves like the reference's (restates
/root/reference/tests/test_vcf_to_h5.py:12-66 against this build's class) and the CLI keeps the six
options of /root/reference/src/haplohyped/vcf_to_h5.py:209-216."""
import os
import tempfile

import pytest

from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter, main


def test_init():
    with tempfile.TemporaryDirectory() as tmpdir:
        sample_file = os.path.join(tmpdir, "samples.txt")
        with open(sample_file, "w") as f:
            f.write("sample1\nsample2\n")
        converter = VCFtoHDF5Converter(cohort_name="test_cohort", vcf_dir="/path/to/vcf", out_dir=tmpdir,
                                       sample_list_path=sample_file, cores=2, cxx_threads=1)
        assert converter.cohort_name == "test_cohort"
        assert converter.cores == 2
        assert converter.cxx_threads == 1
        assert len(converter.donor_ids) == 2
        assert converter.donor_ids == ["sample1", "sample2"]
        assert os.path.exists(converter.tmp_dir)
        assert list(converter.chromosomes) == list(range(1, 23))


def test_read_sample_list(golden_dir):
    with tempfile.TemporaryDirectory() as tmpdir:
        c = VCFtoHDF5Converter("t", "/path/to/vcf", tmpdir, os.path.join(golden_dir, "ipscs_samples_test.txt"), 1, 1)
        assert len(c.donor_ids) == 3 and all(len(d) == 36 for d in c.donor_ids)   # no trailing newline in the fixture


def test_read_sample_list_file_not_found():
    with pytest.raises(FileNotFoundError):
        VCFtoHDF5Converter("test", "/path/to/vcf", "/tmp", "/nonexistent/file.txt", 1, 1)


def test_cli_options():
    names = {p.name: p for p in main.params}
    assert set(names) == {"cohort_name", "vcf", "outdir", "sample_list", "cores", "cxx_threads"}
    assert names["cxx_threads"].default == 4 and names["cohort_name"].required and names["sample_list"].required


def test_facade_surface():
    import parse_vcf
    assert parse_vcf.__doc__ == "Module for parsing VCF files using VCFLoader class"   # parse_vcf.cpp:117
    ld = parse_vcf.VCFLoader()
    assert callable(ld.load_vcf) and callable(ld.load_vcf_without_sample) and callable(parse_vcf.load_vcf)
    import inspect
    assert list(inspect.signature(ld.load_vcf).parameters) == ["in_vcf", "sample", "chrom"]
    assert inspect.signature(ld.load_vcf).parameters["chrom"].default == ""
    with pytest.raises(RuntimeError, match="Error parsing VCF file"):
        ld.load_vcf("/nonexistent.vcf.gz", "x", "chr1")
