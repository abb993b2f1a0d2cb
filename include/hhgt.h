/*
 * hhgt.h — C ABI of libhhgt.so: MI355X (gfx950) genotype encode + Blosc2 shuffle/LZ4 chunk compress.
 *
 * This is the drop-in boundary for ONE hot path of Jaureguy760/HaploHyped-VarAwareML:
 *
 *   reference interface replaced                                       entry point here
 *   ------------------------------------------------------------------ -------------------------
 *   VCFLoader::load_vcf            cpp/parse_vcf.cpp:30-71  (pybind11    hhgt_encode_text (+ hhgt_reader_* for
 *     binding :116-124; per-record GT -> int8 via vcfpp.h:546-588,       files): all samples of a shard in one
 *     isSNP filter vcfpp.h:990-1000)                                     pass -> int8 G[s, v', 2]
 *   VCFLoader::load_vcf_without_sample  cpp/parse_vcf.cpp:80-113        hhgt_encode_text with n_samples = 0
 *   h5py create_dataset(compression=32001,                              hhgt_compress_chunks (byte-shuffle + LZ4
 *     compression_opts=(2,2,0,0,5,1,2))  src/haplohyped/vcf_to_h5.py:134-135   -> Blosc-framed chunks), and
 *     = HDF5 filter 32001 = the Blosc (v1) filter of hdf5-blosc /       hhgt_decompress_chunks (read side,
 *       hdf5plugin.Blosc: shuffle + LZ4-format blocks in Blosc-1         src/utils/h5_reader.py:37-41).  Both the
 *       chunks (Blosc2's filter id is 32026)                Blosc-1 and the Blosc2 header are emitted.
 *   htslib bgzf_read_block + inflate under the reader                  hhgt_bgzf_scan + hhgt_inflate_members
 *     cpp/vcfpp.h:1381 (open), :1468 (record reads)                      (opt-in; the default keeps BGZF on the
 *                                                                        host: include/hhgt_reader.h)
 *
 * Conventions: plain pointers and sizes only.  Pointers named d_* are DEVICE pointers (HBM) on the
 * context's device; everything else is host memory.  `stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  Every function returns HHGT_OK (0) or a negative HHGT_ERR_* code and
 * never aborts the process; hhgt_last_error() returns a thread-local message for the last failure
 * (the reference raises std::runtime_error -> Python RuntimeError, cpp/parse_vcf.cpp:63-66).
 * There is no CPU fallback: without a usable HIP device every compute entry point fails with
 * HHGT_ERR_NO_DEVICE.
 */
#ifndef HHGT_H
#define HHGT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HHGT_OK 0
#define HHGT_ERR_ARG (-1)          /* bad argument                                              */
#define HHGT_ERR_HIP (-2)          /* a HIP runtime call failed                                 */
#define HHGT_ERR_CAPACITY (-3)     /* an output buffer is too small                             */
#define HHGT_ERR_MALFORMED (-4)    /* input text violates the VCF structure the path relies on  */
#define HHGT_ERR_LINE_DENSITY (-5) /* > 1 newline per 16 bytes on average inside a 16 KiB region */
#define HHGT_ERR_NO_DEVICE (-6)    /* no HIP device / kernels for this device                   */
#define HHGT_ERR_IO (-7)           /* file open/read/inflate failure                            */

#define HHGT_BLOSC1 1 /* 16-byte header (c-blosc 1.x), readable by c-blosc2 as well            */
#define HHGT_BLOSC2 2 /* 32-byte extended header (c-blosc2 2.x chunk)                          */

typedef struct hhgt_ctx hhgt_ctx;

const char *hhgt_version(void);
const char *hhgt_last_error(void);
int hhgt_device_count(void);

/* One context per (process, GPU): owns workspaces and timing events.  Not thread-safe; use one
 * context per host thread (the reference's loader object is stateless, cpp/parse_vcf.cpp:19). */
int hhgt_ctx_create(int device, hhgt_ctx **out);
void hhgt_ctx_destroy(hhgt_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * Genotype matrix layout in HBM ("chunk-tiled"):
 *     G[vcol][scol][Sc][Vc][2] int8,  vcol = v / Vc, scol = s / Sc
 * so every HDF5 chunk (Sc samples x Vc variants x 2 haplotypes, C order) is one contiguous
 * Sc*Vc*2-byte run and one sample row of a chunk (Vc*2 bytes) is one Blosc2 block.
 * sc == 0 / vc == 0 select the dense layout G[S][v_capacity][2] (a single chunk).
 * Constraints: vc % 128 == 0; v_capacity % vc == 0; sc is a power of two (or 0).
 * Parity definition: G[s, :, 0] == [t[5] for t in load_vcf(vcf, sample_s, chrom)],
 * G[s, :, 1] == [t[6] ...]  (cpp/parse_vcf.cpp:51-61).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_samples;   /* S: number of sample columns in the VCF (0 = sites only)             */
    int32_t sc;          /* samples per chunk, 0 = dense                                        */
    int32_t vc;          /* variants per chunk, 0 = dense                                       */
    int32_t ring;        /* 0 = linear; > 0: G (and the per-record tables) are a RING of `ring` chunk
                            columns, v_capacity == ring * vc, kept indices are unbounded and wrap
                            (streaming ingest: completed columns leave while later text arrives)  */
    uint64_t v_capacity; /* kept-variant capacity of the G buffer                               */
} hhgt_layout;

/* bytes needed for G under `lay` */
uint64_t hhgt_layout_bytes(const hhgt_layout *lay);
/* byte offset of (s, v, h=0) inside G */
uint64_t hhgt_layout_offset(const hhgt_layout *lay, uint32_t s, uint64_t v);

typedef struct {
    uint64_t n_lines;          /* lines seen (header and blank included)                        */
    uint64_t n_records;        /* data lines                                                    */
    uint64_t n_kept;           /* records kept (region + isSNP)                                 */
    uint64_t n_drop_region;    /* CHROM / range mismatch  (cpp/vcfpp.h:1424-1451)               */
    uint64_t n_drop_filter;    /* !isSNP                  (cpp/vcfpp.h:990-1000)                */
    uint64_t n_haploid_padded; /* calls with one allele: 2nd allele defined as -9               */
    uint64_t n_malformed;      /* records the reference's parser would reject                   */
    uint64_t n_general_lines;  /* kept lines that took the variable-width path                  */
    uint64_t n_chrom_runs;     /* maximal runs of equal CHROM among data lines                  */
} hhgt_encode_stats;

/*
 * Encode one device-resident block of VCF text (whole lines; header lines allowed and skipped).
 *   d_text/nbytes : raw text, 16-byte aligned base, nbytes < 4 GiB.  Only [0, nbytes) is read.
 *   region        : "" / NULL = all records; "chr22"; "chr22:beg-end" (1-based, inclusive)
 *   v_base        : kept records are written at global kept index v_base + k
 *   d_G           : chunk-tiled matrix (see above); d_start/d_stop/d_ref/d_alt: per kept record
 *                   (0-based start, stop = start + len(REF) per cpp/vcfpp.h:1118-1127; REF/ALT are
 *                   single bytes for every kept record by cpp/vcfpp.h:990-1000); any may be NULL.
 *   stats         : host struct, filled on return.
 * Synchronises `stream` before returning (counts are read back).  On HHGT_ERR_MALFORMED the
 * outputs are undefined (the reference raises RuntimeError / asserts).
 * Records of >= 760 samples: the line index does not look at every byte (hhgt_set_index_mode); a pass that comes back
 * MALFORMED is run once more with every byte scanned before this call reports anything, so the outcome is the plain
 * scan's (round 4; the asynchronous forms below make one pass and report it).
 */
int hhgt_encode_text(hhgt_ctx *ctx, const void *d_text, uint64_t nbytes, const char *region,
                     const hhgt_layout *lay, uint64_t v_base, void *d_G, uint32_t *d_start,
                     uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, hhgt_encode_stats *stats,
                     void *stream);

/*
 * Asynchronous form: nothing is read back, so a chain of calls (and what follows them on the stream) can be queued
 * without the host waiting — the streaming ingest (hhgt_ingest.h) and bench.py's step are built on it.
 *   d_cursor  : device uint64.  In: the global kept index the first kept record of this block gets (what v_base is
 *               above).  Out: advanced by the number of kept records.  Chained calls append.
 *   max_lines : caller's bound on the number of lines in the block (a VCF with S samples has at least 2*S + 16 bytes
 *               per data line).  Sizes the workspaces and the grids; lines beyond it are not encoded and reported in
 *               h_result->n_lines_over (hhgt_encode_result_status: HHGT_ERR_CAPACITY).
 *   h_result  : PINNED host memory (hipHostMalloc / torch pin_memory), written by a copy queued on `stream` behind the
 *               kernels; valid once the stream (or an event recorded after the call) has completed.  `done` is set
 *               to 1 by that copy.  May be NULL.
 * Errors that need the counts (malformed records, capacity, line density) are reported by hhgt_encode_result_status
 * on the completed h_result, not by the call.
 */
#define HHGT_RESULT_RUNS 16
typedef struct {
    hhgt_encode_stats stats;
    uint64_t cursor_before, cursor_after;
    uint64_t n_lines_over;     /* lines beyond max_lines (not encoded)                                     */
    uint64_t err_density;      /* newline-slot overflows (HHGT_ERR_LINE_DENSITY)                           */
    uint64_t run_first[HHGT_RESULT_RUNS];   /* CHROM runs of the block: first kept index (block-local) ...  */
    char run_names[HHGT_RESULT_RUNS][32];   /* ... and name; stats.n_chrom_runs may exceed HHGT_RESULT_RUNS */
    uint64_t v_capacity;       /* of the layout the call was given (0 when it was a ring)                  */
    uint32_t done, reserved;
} hhgt_encode_result;

int hhgt_encode_text_async(hhgt_ctx *ctx, const void *d_text, uint64_t nbytes, const char *region,
                           const hhgt_layout *lay, uint64_t *d_cursor, uint32_t max_lines, void *d_G,
                           uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt,
                           hhgt_encode_result *h_result, void *stream);
/* HHGT_OK, or the error the synchronous call would have returned (message via hhgt_last_error) */
int hhgt_encode_result_status(const hhgt_encode_result *r);

/* CHROM runs of the most recent hhgt_encode_text call (cpp/vcfpp.h:1076-1079 CHROM()):
 * run r covers kept indices [first_kept[r], first_kept[r+1]) (batch-local, i.e. without v_base)
 * and is named names[r*32 .. r*32+31] (NUL padded, truncated to 31 bytes). */
int hhgt_encode_chrom_runs(hhgt_ctx *ctx, uint32_t max_runs, uint64_t *first_kept, char *names,
                           uint32_t *n_runs);

/* Zero the padding of G: variants [v_end, round_up(v_end, vc)) of the last chunk column and the
 * sample rows [S, round_up(S, sc)) of chunk columns [vcol_begin, vcol_end). */
int hhgt_pad_tail(hhgt_ctx *ctx, const hhgt_layout *lay, uint64_t v_end, uint64_t vcol_begin,
                  uint64_t vcol_end, void *d_G, void *stream);
/* The same for the chunk column that holds *d_cursor (device uint64, see hhgt_encode_text_async): variants
 * [*d_cursor, round_up(*d_cursor, vc)) and the sample padding rows of that column.  No host round trip. */
int hhgt_pad_tail_cursor(hhgt_ctx *ctx, const hhgt_layout *lay, const uint64_t *d_cursor, void *d_G, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Blosc2 chunk compress: byte-shuffle(typesize) + LZ4 block format, `blocksize`-byte blocks,
 * split into `typesize` streams when 2 <= typesize <= 16 and blocksize/typesize >= 128
 * (flag 0x10 "don't split" is set otherwise).  Replaces the HDF5 filter-32001 call at
 * src/haplohyped/vcf_to_h5.py:134-135 (clevel 5, shuffle 1, LZ4-format codec).
 *   d_src        : n_chunks contiguous chunks of chunk_nbytes bytes each
 *   d_dst        : receives the framed chunks back to back; chunk i occupies
 *                  [d_chunk_off[i], d_chunk_off[i+1])
 *   d_chunk_off  : device array of n_chunks + 1 uint64
 *   total_bytes  : host, optional; when non-NULL the call synchronises and returns the total (and
 *                  HHGT_ERR_CAPACITY if it exceeds dst_cap).  When NULL the call is asynchronous and cannot report an
 *                  overflow afterwards, so it REQUIRES dst_cap >= hhgt_compress_bound(...) (HHGT_ERR_CAPACITY otherwise).
 * Constraints: 1 <= typesize <= 255; blocksize % typesize == 0; 16 <= blocksize <= 65536 (clamped to the chunk size, as c-blosc does);
 * chunk_nbytes < 2 GiB.
 * The streams are valid LZ4 blocks for any input, but the match search is tuned to this path's data.  The case the path is
 * built for — typesize 2, 8 KiB blocks: two 4 KiB byte planes per block, haplotypes of 0 / 1 — is coded from the list of a
 * plane's ones (csrc/lz4bits.hip); planes with other byte values (missing calls, third alleles), very dense planes and
 * every other geometry go through the byte-wise coder (csrc/lz4.hip: minimum match 6, 12-byte hash context, offset-1 run
 * candidate).
 * format: HHGT_BLOSC1 = the 16-byte-header chunk of c-blosc 1.x, what HDF5 filter 32001 stores; HHGT_BLOSC2 = the
 * 32-byte extended header of c-blosc2.
 * ------------------------------------------------------------------------------------------- */
uint64_t hhgt_compress_bound(uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize);
/* Blosc clevel analogue (the reference passes clevel 5: compression_opts[4], vcf_to_h5.py:135) = search effort, candidates
 * tried per position: 1..2: none (offset-1 runs only), 3..4: 1, 5..6: 2 (default 5), 7: 4, 8: 8, 9: 12 and a one-step lazy
 * parse; on 1000G-shaped planes ratio 3.31 / 6.05 / 6.42 / 6.67 / 6.81 / 6.92 at about 0.45 ms more per 3 M x 2504 cohort and
 * candidate (LZ4HC level 5, what the reference's setting selects: 7.01).  Every level emits the same LZ4 block format.
 * The file-writing paths above this header (pipeline.stream_files, the converter) run at 9: they are bound by their input. */
int hhgt_set_clevel(hhgt_ctx *ctx, int clevel);
/* Makes the context's workspaces for the largest encode call (text_bytes, max_lines) and the largest compress call (n_chunks
 * chunks of chunk_nbytes, typesize, blocksize) to come, instead of letting them grow inside the first calls (round 4: what
 * hhgt_ingest_open does when its caller names the sample count).  Either half may be 0.  No counterpart in the reference. */
int hhgt_reserve(hhgt_ctx *ctx, uint64_t text_bytes, uint32_t max_lines, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize,
                 int blocksize);
/* NON-REFERENCE mode (measured separately and labelled so): with on != 0 the record filter also
 * keeps multi-allelic SNP sites — |REF| = 1 and ALT a comma-separated list of single bases from {A,C,G,T} — where the
 * reference's isSNP (cpp/vcfpp.h:990-1000) drops every record with more than two alleles.  Genotypes then carry the
 * allele index as int8 (cpp/vcfpp.h:574), alt[] holds the first ALT base.  Default 0: the reference's filter. */
int hhgt_set_keep_multiallelic(hhgt_ctx *ctx, int on);
/* How the line index of hhgt_encode_text* finds the newlines of records of >= 760 samples (tbx_itr_next's job,
 * cpp/vcfpp.h:1455-1484; results are identical in every mode, tests/test_gpu_index_walk.py):
 *   2 (default) the walk: the head of each line says where its sample columns start; with FORMAT == "GT" the newline of a
 *     record of S diploid calls is at soff + 4 S - 1, and when that byte is one the pass has read 1 KiB of the line;
 *   1 the hop by the bound 2 S + 17 and a search behind it (round 2); 0 the plain scan of every byte; -1 back to the
 *   default (HHGT_INDEX_MODE in the environment, else 2). */
int hhgt_set_index_mode(hhgt_ctx *ctx, int mode);
int hhgt_compress_chunks(hhgt_ctx *ctx, const void *d_src, uint64_t n_chunks, uint64_t chunk_nbytes,
                         int typesize, int blocksize, int format, void *d_dst, uint64_t dst_cap,
                         uint64_t *d_chunk_off, uint64_t *total_bytes, void *stream);

/* Inverse (read side, src/utils/h5_reader.py:37-41): decodes n_chunks framed chunks (either header
 * format) into d_dst, chunk i at d_dst + i*chunk_nbytes.  typesize/blocksize are the dataset's
 * (the headers are validated against them).  Sets *n_bad (host, optional, syncs) to
 * the number of chunks that failed validation. */
int hhgt_decompress_chunks(hhgt_ctx *ctx, const void *d_src, const uint64_t *d_chunk_off,
                           uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize,
                           void *d_dst, uint64_t *n_bad, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Bit-plane form of the genotype matrix: the intermediate between encode and compress when the compressor is the only
 * consumer of the matrix (converter, ingest engine, bench).  Same path, same results — the int8 values
 * cpp/parse_vcf.cpp:46-61 produces and the chunks src/haplohyped/vcf_to_h5.py:131-135 stores — but the 2*S bytes per
 * variant between the two halves shrink to S/2: the encoder writes, and the LZ4 coder reads, two BITS per allele.
 *
 * Geometry: typesize 2, Blosc blocks of 8192 bytes (= one sample x 4096 variants x 2 haplotypes), so the layout needs
 * vc % 4096 == 0 (dense: v_capacity % 4096 == 0).  The planes are TILE-MAJOR: a chunk column (vc variants x S_pad =
 * round_up(S, sc) sample rows; rings: a column slot) is cut into tiles of 256 variants, and tile t of column slot c keeps,
 * for each of four kind-planes kp and each sample row r, one 32-byte piece at
 *     d_P + ((((c * (vc / 256) + t) * 4 + kp) * S_pad + r) * 32        bit i (little-endian dwords, LSB first) = variant
 *                                                                        256 t + i of the column
 *     kp 0: ONE bits, haplotype 0    kp 1: ONE bits, haplotype 1    kp 2: EXC bits, haplotype 0    kp 3: EXC bits, haplotype 1
 * (so an encoder workgroup writes contiguous 8 KiB runs, and a compressor wave gathers the 16 pieces of its plane).
 * (ONE, EXC) = (0,0): allele 0; (1,0): allele 1; (1,1): missing, -9 (cpp/vcfpp.h:567-573); (0,1): any other value
 * (allele index >= 2, cpp/vcfpp.h:574) — its int8 byte sits at the call's ordinary position in d_G (same layout), which
 * is written there and nowhere else.  hhgt_encode_result.reserved counts those calls (exactly — every record is decoded by the variable-width kernel at most once —, saturating at 2^32 - 1); d_G may be NULL where
 * the caller knows there are none (biallelic input under the reference's filter has none).
 * hhgt_planes_bytes(lay) = hhgt_layout_bytes(lay) / 4.
 *
 * hhgt_encode_text_planes_async / hhgt_pad_tail_planes_cursor / hhgt_pad_tail_planes: as their int8 namesakes.
 * hhgt_compress_planes: hhgt_compress_chunks (typesize 2, blocksize 8192) on the chunks of column slots
 *   [col0, col0 + n_cols) of the matrix the planes stand for: n_cols * ceil(S / sc) chunks of sc * vc * 2 bytes, in the
 *   int8 matrix's chunk order; d_P and d_G are the bases of the whole buffers.
 * hhgt_planes_expand: planes (+ d_G's bytes for the (0,1) calls) of those column slots -> int8 bytes at their place in
 *   d_out, a buffer of the int8 layout (d_out == d_G is allowed).  For consumers that want the matrix after all, and for
 *   the parity tests.
 * ------------------------------------------------------------------------------------------- */
uint64_t hhgt_planes_bytes(const hhgt_layout *lay);   /* 0 if the layout cannot carry planes */
int hhgt_encode_text_planes_async(hhgt_ctx *ctx, const void *d_text, uint64_t nbytes, const char *region,
                                  const hhgt_layout *lay, uint64_t *d_cursor, uint32_t max_lines, void *d_P, void *d_G,
                                  uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt,
                                  hhgt_encode_result *h_result, void *stream);
int hhgt_pad_tail_planes_cursor(hhgt_ctx *ctx, const hhgt_layout *lay, const uint64_t *d_cursor, void *d_P, void *stream);
int hhgt_pad_tail_planes(hhgt_ctx *ctx, const hhgt_layout *lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end,
                         void *d_P, void *stream);
int hhgt_compress_planes(hhgt_ctx *ctx, const hhgt_layout *lay, const void *d_P, const void *d_G, uint32_t col0, uint32_t n_cols,
                         int format, void *d_dst, uint64_t dst_cap, uint64_t *d_chunk_off, uint64_t *total_bytes,
                         void *stream);
int hhgt_planes_expand(hhgt_ctx *ctx, const hhgt_layout *lay, const void *d_P, const void *d_G, uint32_t col0, uint32_t n_cols,
                       void *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Consumer side (BASELINE config 5): one-hot haplotype windows straight into a device tensor.
 * Replaces the per-item numpy work of RandomHaplotypeDataset.__getitem__ / encode_haplotypes /
 * encode_sequence (/root/reference/src/datasets/haplotype_dataset.py:54-110,
 * src/utils/common_utils.py:84-103): reference bases of the window, overlaid at every variant of the
 * window with ALT where the donor's allele == 1 and with the VCF REF base otherwise (:99-100), then
 * one-hot float32 [n_items, seq_len, n_channels] for both haplotypes.
 * All pointers inside hhgt_window are DEVICE addresses.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t ref_ptr;       /* uint8 bases (ASCII) of the contig                                   */
    uint64_t ref_len;       /* bases available; window positions beyond it read as 'N'             */
    int64_t win_start;      /* 0-based genomic start of the window                                 */
    uint64_t var_start_ptr; /* uint32 start[] of the group's kept variants (sorted, 0-based)       */
    uint64_t var_ref_ptr;   /* uint8 ref[]                                                         */
    uint64_t var_alt_ptr;   /* uint8 alt[]                                                         */
    uint64_t geno_ptr;      /* int8 pairs (h0,h1) of THIS donor; element 0 belongs to variant geno_first */
    uint32_t geno_first;
    uint32_t var_lo, var_hi; /* variants with win_start <= start < win_start + seq_len             */
    uint32_t reserved;
} hhgt_window;

/* lut: 256 host bytes, base byte -> channel index (0..n_channels-1) or 255 for "no channel"
 * (all-zero row).  d_hap1/d_hap2: float32 [n_items][seq_len][n_channels]. */
int hhgt_onehot_windows(hhgt_ctx *ctx, const hhgt_window *d_items, uint32_t n_items, uint32_t seq_len,
                        const uint8_t *lut, int n_channels, float *d_hap1, float *d_hap2, void *stream);

/* One-hot of a base string for the reference-genome store: replaces
 * ReferenceGenome.array_to_onehot / encode_sequence, /root/reference/src/haplohyped/fasta_encoder.py:47-78
 * (upper-case, non-ACGT -> N, one uint8 column per base, columns in sorted order A,C,G,N,T there).
 * d_out: uint8 [n][n_channels]; lut as in hhgt_onehot_windows. */
int hhgt_onehot_bases_u8(hhgt_ctx *ctx, const uint8_t *d_bases, uint64_t n, const uint8_t *lut, int n_channels,
                         uint8_t *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * BGZF on the device (the north star itself leaves BGZF inflate on the host, where
 * hhgt_reader_* does it with zlib).  Replaces the bgzf_read_block + inflate step htslib runs under the
 * reference's reader (/root/reference/cpp/vcfpp.h:1381 open, :1468 read): the host only walks the member
 * headers, the compressed bytes cross PCIe and every member (<= 64 KiB of text) is inflated by one wave.
 *
 * hhgt_bgzf_scan (host): member table of host[0, nbytes).  Fills up to max_members entries —
 * comp_off (byte offset of the raw DEFLATE payload), comp_len (its length), isize and crc32 (inflated size and
 * CRC-32 from the trailer) — and sets *n_members and *consumed (bytes covered by whole members; a member cut off by the
 * end of the buffer is left for the next call).  HHGT_ERR_MALFORMED if a header is not a BGZF member.
 *
 * hhgt_inflate_members (device): inflates member i from d_src + d_comp_off[i] to d_dst + d_out_off[i]
 * (d_out_off = exclusive prefix sum of isize, computed by the caller).  d_src must be 4-byte aligned and
 * src_bytes a multiple of 4 (pad the upload).  d_status[i] = 0 on success, else a non-zero code
 * (1 block type, 2 stored block, 3 code table, 4 invalid code, 5 distance, 6 output overrun, 7 input overrun,
 * 8 size != ISIZE, 9 CRC-32 of the text != d_crc32[i] — checked by a second kernel when d_crc32 is given, as htslib's
 * bgzf.c does).  *n_bad (host, optional, syncs) = number of members with a non-zero status.
 * ------------------------------------------------------------------------------------------- */
int hhgt_bgzf_scan(const void *host, uint64_t nbytes, uint64_t max_members, uint64_t *comp_off, uint32_t *comp_len,
                   uint32_t *isize, uint32_t *crc32 /* optional */, uint64_t *n_members, uint64_t *consumed);
int hhgt_inflate_members(hhgt_ctx *ctx, const void *d_src, uint64_t src_bytes, const uint64_t *d_comp_off,
                         const uint32_t *d_comp_len, const uint64_t *d_out_off, const uint32_t *d_isize,
                         uint64_t n_members, void *d_dst, uint64_t dst_bytes, const uint32_t *d_crc32 /* optional */,
                         uint32_t *d_status, uint64_t *n_bad, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-stage device timing (HIP events on the launch stream).  Stages are indexed by HHGT_STAGE_*.
 * hhgt_profile_read returns accumulated milliseconds and launch counts since the last reset.
 * ------------------------------------------------------------------------------------------- */
#define HHGT_STAGE_INDEX 0    /* newline index + compaction                                     */
#define HHGT_STAGE_FIXED 1    /* fixed-column parse, filter, kept-record compaction              */
#define HHGT_STAGE_ENCODE 2   /* GT tile encode (fixed-width path)                               */
#define HHGT_STAGE_GENERAL 3  /* GT encode, variable-width lines                                 */
#define HHGT_STAGE_LZ4 4      /* shuffle + LZ4 block encode                                      */
#define HHGT_STAGE_FRAME 5    /* Blosc2 framing / compaction                                     */
#define HHGT_STAGE_DECODE 6   /* chunk decode                                                    */
#define HHGT_STAGE_ONEHOT 7   /* one-hot haplotype windows                                       */
#define HHGT_STAGE_INFLATE 8  /* BGZF members inflated on the device                            */
#define HHGT_N_STAGES 9
int hhgt_profile_enable(hhgt_ctx *ctx, int on);
int hhgt_profile_reset(hhgt_ctx *ctx);
int hhgt_profile_read(hhgt_ctx *ctx, double *ms /*[HHGT_N_STAGES]*/, uint64_t *launches /*[HHGT_N_STAGES]*/);

/* ---------------------------------------------------------------------------------------------
 * Streams for callers that run the compress step of one block BESIDE hhgt_encode_text*_async of the next (two
 * streams, events in between; what bench.py does).  Both are plain hipStream_t values, usable wherever this header
 * takes `void *stream`.
 *   HHGT_STREAM_ENCODE    high priority, every CU
 *   HHGT_STREAM_COMPRESS  default priority, restricted (hipExtStreamCreateWithCUMask) to 3/4 of the CUs: the LZ4 kernel
 *                         is bound by instruction issue and fills every CU it may use with resident waves, and the
 *                         encode chain of the next block — two HBM-bound passes with latency-bound small kernels in
 *                         between — then waits for slots one workgroup at a time (k_parse_fixed: 146 us beside LZ4, 40 us
 *                         alone).  HHGT_COMPRESS_CUS=n overrides the CU count (0 = all).
 *   HHGT_STREAM_FRAME     default priority, every CU: for hhgt_set_frame_stream
 * Destroy with hhgt_stream_destroy before hhgt_ctx_destroy.
 * ------------------------------------------------------------------------------------------- */
#define HHGT_STREAM_ENCODE 0
#define HHGT_STREAM_COMPRESS 1
#define HHGT_STREAM_FRAME 2
int hhgt_stream_create(hhgt_ctx *ctx, int kind, void **stream);
int hhgt_stream_destroy(hhgt_ctx *ctx, void *stream);
/* A third stream for the framing half of hhgt_compress_chunks / hhgt_compress_planes (round 4).  The LZ4 kernels of a call
 * still run on the stream the call names; header, bstarts and the copy of the streams to their final place (k_frame_*, the
 * filter-32001 chunk layout of vcf_to_h5.py:134-135) are queued on `stream` behind them, so that the caller's stream can
 * go on with the LZ4 kernels of the NEXT call (the codec workspaces are double-buffered; a call waits for the framing
 * that last read its set).  d_dst / d_chunk_off of a call are complete in `stream` order: record events THERE.  With
 * total_bytes != NULL (the synchronous form) the call still returns with the chunks complete.  NULL: back to one stream.
 * The stream must outlive every compress call that uses it (hhgt_stream_destroy on it resets this first). */
int hhgt_set_frame_stream(hhgt_ctx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif
