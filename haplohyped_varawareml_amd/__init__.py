"""haplohyped_varawareml_amd — MI355X-native genotype encode + Blosc2 shuffle/LZ4 compress path.

Scope: the VCF GT-text -> int8 genotype matrix encode and the Blosc2 chunk
compress loop of Jaureguy760/HaploHyped-VarAwareML, as hand-written HIP kernels for gfx950 behind
a C ABI (include/hhgt.h, libhhgt.so), with the reference's own Python surfaces on top
(`parse_vcf.VCFLoader`, `vcf_to_h5`).  Importing this package does not load the native library;
the first use does, and raises if it has not been built.
"""
__version__ = "0.1.0"
