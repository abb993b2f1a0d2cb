/*
 * hhgt_reader.h — host side of the path: VCF text from disk into pinned memory, ready for the device.
 *
 * Replaces what the reference gets from htslib through vcfpp (NOT in the reference tree; unpinned
 * system dependency, environment.yml:16):
 *   hts_open / bcf_hdr_read ............ /root/reference/cpp/vcfpp.h:1378-1385
 *   tbx_itr_next -> bgzf_getline ....... /root/reference/cpp/vcfpp.h:1468   (BGZF inflate + line framing)
 * As BASELINE.json's north star prescribes, decompression stays on the host cores: BGZF members are
 * inflated by a pool of worker threads that is never stopped at block boundaries (the scanner lays out later
 * blocks while earlier ones are still being inflated), plain gzip (like the reference's fixture
 * tests/data/chr22.filtered.vcf.gz) by one streaming inflater; the text lands in a ring of PINNED
 * buffers cut at line boundaries, from which hhgt_reader_copy_async issues hipMemcpyAsync on the
 * caller's stream (double buffering: the copy of block k+1 overlaps the kernels of block k).
 * No tabix index is needed: region selection happens on the device (hhgt_encode_text's `region`).
 * Every BGZF member's text is hashed (hhgt_crc32) and compared with the member's trailer, as htslib's
 * bgzf.c does; a mismatch fails the read with "CRC32 checksum mismatch" (environment
 * HHGT_BGZF_NO_CRC=1 skips the check).
 */
#ifndef HHGT_READER_H
#define HHGT_READER_H
#include "hhgt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hhgt_reader hhgt_reader;

/* path: .vcf, .vcf.gz (gzip or BGZF).  block_bytes: size of each pinned block (>= 1 MiB; lines longer
 * than a block are an error).  n_threads: BGZF inflate workers (0 = hardware concurrency, max 192).
 * n_blocks: ring depth (>= 2; 0 = 4). */
int hhgt_reader_open(const char *path, uint64_t block_bytes, int n_threads, int n_blocks, hhgt_reader **out);
void hhgt_reader_close(hhgt_reader *r);

/* 1 if the file is BGZF (block-parallel inflate), 0 for plain gzip / uncompressed */
int hhgt_reader_is_bgzf(const hhgt_reader *r);

/* Next block of whole lines.  *host_ptr stays valid until the next hhgt_reader_next/close call.
 * *nbytes == 0 at end of file.  The final line is delivered even without a trailing newline. */
int hhgt_reader_next(hhgt_reader *r, const void **host_ptr, uint64_t *nbytes);

/* The same with explicit ownership, for consumers that keep several blocks in flight (the ingest engine uploads
 * block k+1 while block k is still being copied): the block stays valid until hhgt_reader_release(token).
 * *is_last (optional) = 1 for the final block of the file.  End of file: *nbytes == 0, *token == -1. */
int hhgt_reader_acquire(hhgt_reader *r, const void **host_ptr, uint64_t *nbytes, int *token, int *is_last);
int hhgt_reader_release(hhgt_reader *r, int token);

/* CPUs this process may use: affinity mask capped by the cgroup CPU quota (the reader's default thread count) */
int hhgt_effective_cpus(void);

/* Closed readers keep up to 32 pinned ring blocks for the next hhgt_reader_open (pinning memory is slow and a
 * converter opens one reader per chromosome file); this frees them. */
void hhgt_reader_trim_pool(void);
/* ... and this puts up to n_blocks pinned blocks of block_bytes there ahead of time (the ingest engine does, when its caller
 * names the sample count at hhgt_ingest_open: the first file then finds its ring pinned).  Returns how many the pool holds
 * of that size afterwards; without a HIP device nothing is made. */
int hhgt_reader_prewarm(uint64_t block_bytes, int n_blocks);

/* CRC-32 (RFC 1952) of a host buffer: the carry-less-multiply folding form where the CPU has PCLMULQDQ (what the
 * reader checks every BGZF member with), the 16-byte table form elsewhere. */
uint32_t hhgt_crc32(const void *data, uint64_t n);

/* Raw DEFLATE (RFC 1951) of one BGZF member: 0 iff the stream is complete and inflates to exactly out_len bytes.  The
 * decoder the reader's workers use (csrc/fast_inflate.h) before falling back to zlib; never touches memory outside
 * the two buffers. */
int hhgt_fast_inflate(const void *in, uint64_t in_len, void *out, uint64_t out_len);

/* hipMemcpyAsync(d_dst, host_ptr, nbytes, HostToDevice, stream) from the reader's pinned block */
int hhgt_reader_copy_async(hhgt_reader *r, const void *host_ptr, uint64_t nbytes, void *d_dst, void *stream);

/* totals so far: compressed bytes consumed from disk, text bytes produced */
int hhgt_reader_stats(const hhgt_reader *r, uint64_t *file_bytes, uint64_t *text_bytes);

#ifdef __cplusplus
}
#endif
#endif
