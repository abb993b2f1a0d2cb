"""Top-level module named like the reference's pybind11 extension (`import parse_vcf`,
/root/reference/cpp/parse_vcf.cpp:116): re-exports the MI355X-native implementation."""
from haplohyped_varawareml_amd.parse_vcf import VCFLoader, load_vcf, load_vcf_without_sample  # noqa: F401

__doc__ = "Module for parsing VCF files using VCFLoader class"
