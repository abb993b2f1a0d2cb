"""Packaging shim: the same console scripts the reference installs (/root/reference/setup.py:24-29).
The native library is built in-tree (python -m haplohyped_varawareml_amd.build), not by this file."""
from setuptools import setup

setup(
    name="haplohyped-varawareml-amd",
    version="0.1.0",
    description="MI355X-native VCF genotype encode + Blosc2 compress path (hhgt)",
    packages=["haplohyped_varawareml_amd"],
    py_modules=["parse_vcf"],
    package_data={"haplohyped_varawareml_amd": ["libhhgt.so", "csrc/*"]},
    python_requires=">=3.10",
    entry_points={"console_scripts": [
        "vcf_to_h5=haplohyped_varawareml_amd.vcf_to_h5:main",
        "fasta_encoder=haplohyped_varawareml_amd.fasta_encoder:main",
    ]},
)
