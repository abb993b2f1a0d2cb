/*
 * hhgt_synth.h — synthetic-workload generator entry points of libhhgt.so.
 * BENCH / TEST TOOLING, not part of the reference-facing boundary: the reference ships no generator;
 * BASELINE.md §3 defines the synthetic VCF shapes (1000G-style biallelic phased "a|b" columns).
 * The generator writes VCF TEXT directly into HBM so that 30 GB workloads (3 M variants x 2504
 * samples) never cross PCIe; its byte-exact CPU mirror is haplohyped_varawareml_amd/synth.py.
 */
#ifndef HHGT_SYNTH_H
#define HHGT_SYNTH_H
#include "hhgt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Renders n_variants fixed-width data lines
 *     CHROM \t POS \t . \t REF \t ALT \t . \t PASS \t . \t GT \t a|b \t a|b ... a|b \n
 * at d_text + d_line_off[i].  Allele (v, s, h) = 1 iff
 *     (mix64(key ^ (v_first+i)*0xD1B54A32D192ED03 ^ (2s+h)*0x9E3779B97F4A7C15) >> 32) < d_thr[i],
 * key = mix64(seed + 0x9E3779B97F4A7C15), mix64 = splitmix64 finaliser.
 * d_line_off has n_variants + 1 entries (byte offsets; line i must be exactly
 * strlen(contig) + digits(pos) + 20 + 4*n_samples bytes long). */
int hhgt_synth_render_fixed(hhgt_ctx *ctx, void *d_text, uint64_t text_cap, const uint64_t *d_line_off,
                            const uint32_t *d_pos, const uint8_t *d_ref, const uint8_t *d_alt,
                            const uint32_t *d_thr, uint64_t n_variants, uint64_t v_first, const char *contig,
                            int n_samples, uint64_t seed, void *stream);

/* BASELINE config 4 shape: multiallelic records (ALT "C,G", "C,G,T"), indel / '*' / '<DEL>' / lower-case
 * ALTs that the isSNP filter must drop, missing ('.') and half-missing calls, '/' separators and records
 * with GT:DP columns ("a|b:dd").  Per variant i: d_ref8/d_alt8 hold the REF / ALT text (up to 8 bytes,
 * little-endian packed), d_meta = ref_len | alt_len << 4 | n_alt << 8 | with_dp << 12.
 * Per call (v, s), with r_h = mix64(kv ^ (2s+h)*GOLD) and r2 = mix64(kv ^ (2s)*GOLD ^ 0xA5A5A5A5A5A5A5A5):
 *   allele_h = (r_h >> 32) < thr ? (n_alt > 1 ? 1 + ((r_h >> 8) & 0xFFFFFF) % n_alt : 1) : 0
 *   (r2 & 0xFFFF) < 1311 -> both '.';  < 1639 -> allele ((r2 >> 16) & 1) is '.'
 *   ((r2 >> 20) & 0xFFFF) < 3277 -> separator '/', else '|';   DP = 10 + (r2 >> 40) % 90
 * Line i is strlen(contig) + digits(pos) + ref_len + alt_len + 16 + fmt_len + S * (with_dp ? 7 : 4)
 * bytes (fmt_len = 2 or 5).  CPU mirror: haplohyped_varawareml_amd/synth.py (mixed_*). */
int hhgt_synth_render_mixed(hhgt_ctx *ctx, void *d_text, uint64_t text_cap, const uint64_t *d_line_off,
                            const uint32_t *d_pos, const uint64_t *d_ref8, const uint64_t *d_alt8,
                            const uint32_t *d_meta, const uint32_t *d_thr, uint64_t n_variants, uint64_t v_first,
                            const char *contig, int n_samples, uint64_t seed, void *stream);

/* BGZF writer for synthetic shards (BASELINE.md §3 asks for BGZF-compressed inputs; htslib / bgzip are not in the
 * image): host text -> 0xFF00-byte members deflated at `level` by n_threads threads (0 = hardware concurrency) ->
 * file, with the empty end-of-file member.  Byte-identical to haplohyped_varawareml_amd.reader.write_bgzf. */
int hhgt_synth_write_bgzf(const char *path, const void *text, uint64_t nbytes, int level, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
