/*
 * hhgt_ingest.h — streaming ingest: .vcf / .vcf.gz files (or text already in host memory) -> framed genotype
 * chunks back on the host, with every stage running concurrently and nothing waiting on the host in between.
 *
 * This is the body of VCFtoHDF5Converter.genotype_vcf_to_hdf5 (/root/reference/src/haplohyped/vcf_to_h5.py:79-140:
 * parse_vcf.load_vcf per donor -> re-pack -> create_dataset(filter 32001)) for ALL samples of a file at once, as one
 * native pipeline:
 *
 *   source thread   host mode (the north star's: BGZF inflate stays on the host cores): hhgt_reader_* worker pool ->
 *                   pinned ring -> hipMemcpyAsync into one of the device text buffers;
 *                   device mode (opt-in): member headers walked on the host, the COMPRESSED members
 *                   cross PCIe, hhgt_inflate_members' kernels write the text;
 *                   memory mode: text already in (pinned) host memory is cut at line ends and uploaded (the
 *                   "host-fed" measurement leg).  The next inputs are opened ahead, so their inflate overlaps the
 *                   encode of the current one.
 *   driver thread   hhgt_encode_text_async appends every block at a device-resident cursor into a RING of chunk
 *                   columns; one block later (the GPU already has the next block queued) it reads the block's result
 *                   record, queues the device -> host copy of the new rows of the variant tables and
 *                   hhgt_compress_chunks for the chunk columns the block completed
 *   shipper thread  waits for a batch's sizes, copies the framed bytes into pinned memory, hands the event out
 *   caller          hhgt_ingest_next
 *
 * The results are identical to encoding the whole file with one hhgt_encode_text call followed by
 * hhgt_compress_chunks (tests/test_gpu_ingest.py checks that against the oracle, for every source mode).
 */
#ifndef HHGT_INGEST_H
#define HHGT_INGEST_H
#include "hhgt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hhgt_ingest hhgt_ingest;

typedef struct {
    int32_t sc, vc;         /* chunk geometry: samples / variants per chunk (0 = 64 / 8192)                        */
    int32_t typesize;       /* 0 = 2: one diploid call                                                             */
    int32_t blocksize;      /* Blosc block bytes, 0 = min(vc * 2, 8192)                                            */
    int32_t format;         /* HHGT_BLOSC1 (what HDF5 filter 32001 stores) or HHGT_BLOSC2; 0 = HHGT_BLOSC2          */
    int32_t sites_only;     /* 1: ignore sample columns (load_vcf_without_sample, cpp/parse_vcf.cpp:80-113)         */
    int32_t device_inflate; /* 0: the host reader inflates (the north star's design); 1: BGZF files are inflated on the
                               device; 2 = auto: BGZF files whose first member inflates to >= 2x its size take the
                               device (the compressed members are then the smaller load for the link, which is what
                               bounds the host path at cohort widths); non-BGZF input always takes the host reader  */
    int32_t n_threads;      /* host inflate threads per open file (0 = the reader's default)                       */
    uint64_t block_bytes;   /* text block size; 0 = 64 MiB (host reader / memory) or 512 MiB (device inflate)         */
    int32_t files_ahead;    /* inputs opened ahead of the one being uploaded (host reader), 0 = 1                   */
    int32_t expect_samples; /* > 0: the sample count the inputs will have (their #CHROM line says it; the converter reads
                               the first file's header): hhgt_ingest_open then makes and pins every buffer whose size
                               follows from it — ring, batch slots, pinned copies, staging, the host reader's pinned
                               blocks — instead of the first input (round 3: 9.35 M variants/s in the first pass of the
                               device-inflate leg against 11.1 M steady).  0: sized when the first header arrives.  A
                               wider input still grows what it needs.                                                */
} hhgt_ingest_opts;

typedef struct {
    uint64_t n_samples, n_lines, n_records, n_kept, n_drop_region, n_drop_filter, n_haploid_padded, n_general_lines;
    uint64_t text_bytes, file_bytes, raw_bytes, compressed_bytes, n_blocks;
    double seconds;         /* first block of the input queued on the device .. its last batch delivered           */
    int32_t is_bgzf, device_inflate;
} hhgt_ingest_stats;

#define HHGT_EV_END 0        /* every queued input is done (and hhgt_ingest_finish was called)                     */
#define HHGT_EV_HEADER 1     /* header / n_samples of the input                                                     */
#define HHGT_EV_VARIANTS 2   /* the kept records of one text block: start / ref / alt rows, CHROM runs             */
#define HHGT_EV_COLUMNS 3    /* completed chunk columns, framed                                                    */
#define HHGT_EV_INPUT_END 4  /* stats of the input                                                                  */

typedef struct {
    int32_t kind, input;     /* input: index in the order of the hhgt_ingest_add_* calls                            */
    /* HHGT_EV_HEADER: the '#' lines of the file, verbatim */
    const char *header;
    uint64_t header_bytes, n_samples;
    /* HHGT_EV_VARIANTS: rows [first_variant, first_variant + n_variants) of the input's kept records
     * (0-based start; stop = start + 1 for every kept record, cpp/vcfpp.h:1118-1127 with |REF| = 1) */
    const uint32_t *start;
    const uint8_t *ref, *alt;
    uint64_t first_variant, n_variants;
    uint32_t n_runs, pad_;          /* CHROM runs that BEGIN in this block: first kept index (input-global), name */
    const uint64_t *run_first;
    const char *run_names;          /* 32 bytes each, NUL padded */
    /* HHGT_EV_COLUMNS: chunk columns [first_col, first_col + n_cols); n_chunks = n_cols * ceil(S / sc) chunks,
     * column-major then sample-chunk; chunk i = framed[chunk_off[i], chunk_off[i+1]) */
    const uint8_t *framed;
    const uint64_t *chunk_off;
    uint64_t framed_bytes, n_chunks, first_col, n_cols, raw_bytes;
    /* HHGT_EV_INPUT_END */
    hhgt_ingest_stats stats;
} hhgt_ingest_event;

/* The engine drives `ctx` from its own threads: do not use the context for anything else until hhgt_ingest_close. */
int hhgt_ingest_open(hhgt_ctx *ctx, const hhgt_ingest_opts *opts, hhgt_ingest **out);
/* Queue an input (processed in the order added; may be called while earlier inputs are running).  region as in
 * hhgt_encode_text.  Returns the input's index (>= 0) or a negative error. */
int hhgt_ingest_add_file(hhgt_ingest *g, const char *path, const char *region);
/* Text in host memory (ideally pinned: hipHostMalloc / torch pin_memory), whole VCF text starting with its header.
 * The memory must stay valid until the input's HHGT_EV_INPUT_END. */
int hhgt_ingest_add_memory(hhgt_ingest *g, const void *host_text, uint64_t nbytes, const char *region);
/* No more inputs will be added: HHGT_EV_END follows the last input's events. */
int hhgt_ingest_finish(hhgt_ingest *g);
/* Next event, in order.  Blocks.  Pointers stay valid until the next hhgt_ingest_next / hhgt_ingest_close call.
 * On an error (first failing stage wins) returns its code; the engine is then finished. */
int hhgt_ingest_next(hhgt_ingest *g, hhgt_ingest_event *ev);
/* Keep the buffers of the event hhgt_ingest_next returned last beyond the next call (a consumer that writes a batch of
 * chunks to a file on another thread while it takes the next event): *token names them, hhgt_ingest_release(token) gives
 * them back, from any thread.  The engine has 5 chunk buffers and 6 variant-table buffers: a consumer that holds them all
 * gets no further event until it releases one.  Once per event; an event without buffers gives a token that releases nothing. */
int hhgt_ingest_hold(hhgt_ingest *g, int *token);
int hhgt_ingest_release(hhgt_ingest *g, int token);
void hhgt_ingest_close(hhgt_ingest *g);

#ifdef __cplusplus
}
#endif
#endif
