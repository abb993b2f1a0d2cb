/*
 * hhgt_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's genotype-encode + Blosc2 shuffle/LZ4 path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped product path (haplohyped_varawareml_amd/csrc, libhhgt.so) never links or calls it.
 *
 * Parity pinning status
 *   - GT -> int8 semantics: pinned by the golden vectors derived from the reference's own fixture
 *     (tests/golden/chr22.filtered.vcf.gz -> tests/golden/fixture_golden.json, made by
 *     tests/golden/make_golden.py with an independent pure-Python splitter) and by hand-written
 *     known-answer lines, each tied to a cpp/vcfpp.h line.  The reference C++ path itself is
 *     UNBUILDABLE here (needs htslib, absent) so there is no oracle/_ref.
 *   - LZ4 block format: pinned against system liblz4 1.9.3 (LZ4_decompress_safe / LZ4_compress_default).
 *   - byte-shuffle + Blosc(1) chunk layout: pinned against c-blosc 1.21 (/opt/conda/lib/libblosc.so.1).
 *   - Blosc2 extended header (32 B): restated from the published c-blosc2 chunk format
 *     (hdf5plugin>=4.0 bundles c-blosc2 2.x; neither is in the container)  => "parity unpinned"
 *     for those 16 extra header bytes; everything after the header is the Blosc1-pinned layout.
 */
#ifndef HHGT_ORACLE_H
#define HHGT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t n_lines;          /* all lines seen (incl. header, blank)              */
    int64_t n_records;        /* data lines (not starting with '#', non-empty)     */
    int64_t n_kept;           /* records that passed region + isSNP                */
    int64_t n_drop_region;    /* CHROM / range mismatch                            */
    int64_t n_drop_filter;    /* failed isSNP (cpp/vcfpp.h:990-1000)               */
    int64_t n_haploid_padded; /* calls with a single allele (2nd padded with -9)   */
    int64_t n_malformed;      /* records with < 8 tabs, bad POS, FORMAT w/o GT ... */
} oracle_vcf_stats;

/* Matrix-shaped encode: one pass over `text`, all samples.
 * G is sample-major [n_samples][cap][2] int8 (row stride = cap*2 bytes).
 * start/stop: 0-based start, stop = start + len(REF)   (cpp/vcfpp.h:1118-1127)
 * ref/alt: one byte each (kept records have |REF|==|ALT|==1 by cpp/vcfpp.h:990-1000)
 * chrom: [cap][32] NUL padded CHROM text, may be NULL.
 * region: "" / NULL = no filter; "chrN" ; "chrN:beg-end" (1-based inclusive, tabix style).
 * Returns n_kept, or -1 if cap is too small / -2 on malformed input. */
/* labelled NON-REFERENCE filter mode (include/hhgt.h hhgt_set_keep_multiallelic): multi-allelic SNP sites are kept */
void oracle_set_keep_multiallelic(int on);
int64_t oracle_vcf_encode(const uint8_t *text, size_t n, const char *region, int n_samples,
                          size_t cap, int8_t *G, uint32_t *start, uint32_t *stop,
                          uint8_t *ref, uint8_t *alt, char *chrom, oracle_vcf_stats *st);

/* Reference-shaped: one sample column (index into the header's sample order), re-scanning every
 * line in full like htslib's vcf_parse1 does (cpp/parse_vcf.cpp:37-61).  phase: [cap][2]. */
int64_t oracle_vcf_load_sample(const uint8_t *text, size_t n, const char *region, int n_samples,
                               int sample_index, size_t cap, int8_t *phase, uint32_t *start,
                               uint32_t *stop, uint8_t *ref, uint8_t *alt, oracle_vcf_stats *st);

/* Header: finds the #CHROM line, returns number of samples; writes offsets of the sample names
 * (into text) if name_off/name_len non-NULL (capacity max_names). -1 if no #CHROM line. */
int oracle_vcf_header_samples(const uint8_t *text, size_t n, uint32_t *name_off, uint32_t *name_len,
                              int max_names);

/* ---- codec ---- */
void oracle_shuffle(const uint8_t *src, uint8_t *dst, size_t nbytes, int typesize);
void oracle_unshuffle(const uint8_t *src, uint8_t *dst, size_t nbytes, int typesize);

int oracle_lz4_bound(int n);
/* greedy single-hash LZ4 block compressor (LZ4 block format); returns csize, 0 if it does not fit */
int oracle_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap);
/* safe decoder; returns decoded size or <0 on malformed input */
int oracle_lz4_decompress(const uint8_t *src, int csize, uint8_t *dst, int cap);

#define ORACLE_BLOSC1 1 /* 16-byte header, version 2 (c-blosc 1.x layout)           */
#define ORACLE_BLOSC2 2 /* 32-byte extended header, version 5 (c-blosc2 2.x layout)  */

size_t oracle_blosc_bound(size_t nbytes, int typesize, int blocksize);
/* shuffle + split + LZ4 per stream + framing.  Returns cbytes or <0. */
int64_t oracle_blosc_compress(const uint8_t *src, size_t nbytes, int typesize, int blocksize,
                              int format, uint8_t *dst, size_t cap);
/* decodes either header format; returns nbytes or <0 */
int64_t oracle_blosc_decompress(const uint8_t *chunk, size_t cbytes, uint8_t *dst, size_t cap);
int oracle_blosc_info(const uint8_t *chunk, size_t avail, uint32_t *nbytes, uint32_t *blocksize,
                      uint32_t *cbytes, int *typesize, int *flags, int *version);

#ifdef __cplusplus
}
#endif
#endif
