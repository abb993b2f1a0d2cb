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

#ifdef __cplusplus
}
#endif
#endif
