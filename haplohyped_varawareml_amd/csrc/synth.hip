// synth.hip — synthetic VCF text rendered directly in HBM (bench / test tooling; see
// include/hhgt_synth.h).  One workgroup per line; lanes write 16 B (4 "a|b\t" fields) each.
#include "common.h"
#include "../../include/hhgt_synth.h"
#include <string.h>

__host__ __device__ static inline unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

struct SynthContig {
    char name[32];
    int len;
};

typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));

__global__ __launch_bounds__(256) void k_synth_fixed(uint8_t *__restrict__ text, uint64_t text_cap,
                                                     const unsigned long long *__restrict__ line_off,
                                                     const uint32_t *__restrict__ pos, const uint8_t *__restrict__ ref,
                                                     const uint8_t *__restrict__ alt, const uint32_t *__restrict__ thr,
                                                     uint64_t n_variants, uint64_t v_first, SynthContig contig,
                                                     uint32_t S, unsigned long long key)
{
    for (uint64_t i = blockIdx.x; i < n_variants; i += gridDim.x) {
        const unsigned long long off = line_off[i], end = line_off[i + 1];
        if (end > text_cap) continue;
        const uint32_t p = pos[i];
        uint32_t nd = 1;
        for (uint32_t t = p; t >= 10u; t /= 10u) ++nd;
        const uint32_t plen = (uint32_t)contig.len + 1u + nd + 19u;
        if (threadIdx.x == 0) {
            uint8_t *o = text + off;
            for (int k = 0; k < contig.len; ++k) *o++ = (uint8_t)contig.name[k];
            *o++ = '\t';
            uint32_t t = p;
            for (uint32_t k = 0; k < nd; ++k) {
                o[nd - 1u - k] = (uint8_t)('0' + t % 10u);
                t /= 10u;
            }
            o += nd;
            const char mid[] = "\t.\t";
            for (int k = 0; k < 3; ++k) *o++ = (uint8_t)mid[k];
            *o++ = ref[i];
            *o++ = '\t';
            *o++ = alt[i];
            const char tail[] = "\t.\tPASS\t.\tGT\t";
            for (int k = 0; k < 13; ++k) *o++ = (uint8_t)tail[k];
        }
        const unsigned long long kv = key ^ ((v_first + i) * 0xD1B54A32D192ED03ull);
        const uint32_t th = thr[i];
        uint8_t *sb = text + off + plen;
        for (uint32_t s0 = threadIdx.x * 4u; s0 < S; s0 += blockDim.x * 4u) {
            uint32_t d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t s = s0 + q;
                const uint32_t a = (uint32_t)(mix64(kv ^ ((unsigned long long)(2u * s) * 0x9E3779B97F4A7C15ull)) >> 32) < th;
                const uint32_t b = (uint32_t)(mix64(kv ^ ((unsigned long long)(2u * s + 1u) * 0x9E3779B97F4A7C15ull)) >> 32) < th;
                const uint32_t term = (s == S - 1u) ? (uint32_t)'\n' : (uint32_t)'\t';
                d[q] = ('0' + a) | ((uint32_t)'|' << 8) | (('0' + b) << 16) | (term << 24);
            }
            if (s0 + 4u <= S) {
                u32x4_u v = {d[0], d[1], d[2], d[3]};
                *reinterpret_cast<u32x4_u *>(sb + 4ull * s0) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (s0 + q < S)
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) sb[4ull * (s0 + q) + bb] = (uint8_t)(d[q] >> (8 * bb));
            }
        }
    }
}

extern "C" int hhgt_synth_render_fixed(hhgt_ctx *c, void *d_text, uint64_t text_cap, const uint64_t *d_line_off,
                                       const uint32_t *d_pos, const uint8_t *d_ref, const uint8_t *d_alt,
                                       const uint32_t *d_thr, uint64_t n_variants, uint64_t v_first,
                                       const char *contig, int n_samples, uint64_t seed, void *stream)
{
    if (!c || !d_text || !d_line_off || !d_pos || !d_ref || !d_alt || !d_thr || !contig || n_samples < 1) {
        hhgt_set_error("synth: bad arguments");
        return HHGT_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    SynthContig sc;
    memset(&sc, 0, sizeof(sc));
    size_t l = strlen(contig);
    if (l == 0 || l > 31) {
        hhgt_set_error("synth: contig name must be 1..31 bytes");
        return HHGT_ERR_ARG;
    }
    memcpy(sc.name, contig, l);
    sc.len = (int)l;
    if (n_variants == 0) return HHGT_OK;
    uint32_t grid = n_variants < 65536ull * 4 ? (uint32_t)n_variants : 65536u * 4u;
    hipLaunchKernelGGL(k_synth_fixed, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<uint8_t *>(d_text), text_cap,
                       reinterpret_cast<const unsigned long long *>(d_line_off), d_pos, d_ref, d_alt, d_thr,
                       n_variants, v_first, sc, (uint32_t)n_samples, mix64(seed + 0x9E3779B97F4A7C15ull));
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// ---------------------------------------------------------------------------------------------
// config-4 style lines (see include/hhgt_synth.h)
__global__ __launch_bounds__(256) void k_synth_mixed(uint8_t *__restrict__ text, uint64_t text_cap,
                                                     const unsigned long long *__restrict__ line_off,
                                                     const uint32_t *__restrict__ pos,
                                                     const unsigned long long *__restrict__ ref8,
                                                     const unsigned long long *__restrict__ alt8,
                                                     const uint32_t *__restrict__ meta, const uint32_t *__restrict__ thr,
                                                     uint64_t n_variants, uint64_t v_first, SynthContig contig, uint32_t S,
                                                     unsigned long long key)
{
    for (uint64_t i = blockIdx.x; i < n_variants; i += gridDim.x) {
        const unsigned long long off = line_off[i], end = line_off[i + 1];
        if (end > text_cap) continue;
        const uint32_t p = pos[i], m = meta[i];
        const uint32_t ref_len = m & 15u, alt_len = (m >> 4) & 15u, n_alt = (m >> 8) & 15u, with_dp = (m >> 12) & 1u;
        uint32_t nd = 1;
        for (uint32_t t = p; t >= 10u; t /= 10u) ++nd;
        const uint32_t fmt_len = with_dp ? 5u : 2u;
        const uint32_t plen = (uint32_t)contig.len + 1u + nd + 3u + ref_len + 1u + alt_len + 10u + fmt_len + 1u;
        if (threadIdx.x == 0) {
            uint8_t *o = text + off;
            for (int k = 0; k < contig.len; ++k) *o++ = (uint8_t)contig.name[k];
            *o++ = '\t';
            uint32_t t = p;
            for (uint32_t k = 0; k < nd; ++k) {
                o[nd - 1u - k] = (uint8_t)('0' + t % 10u);
                t /= 10u;
            }
            o += nd;
            *o++ = '\t'; *o++ = '.'; *o++ = '\t';
            for (uint32_t k = 0; k < ref_len; ++k) *o++ = (uint8_t)(ref8[i] >> (8 * k));
            *o++ = '\t';
            for (uint32_t k = 0; k < alt_len; ++k) *o++ = (uint8_t)(alt8[i] >> (8 * k));
            const char tail[] = "\t.\tPASS\t.\t";
            for (int k = 0; k < 10; ++k) *o++ = (uint8_t)tail[k];
            *o++ = 'G'; *o++ = 'T';
            if (with_dp) { *o++ = ':'; *o++ = 'D'; *o++ = 'P'; }
            *o++ = '\t';
        }
        const unsigned long long kv = key ^ ((v_first + i) * 0xD1B54A32D192ED03ull);
        const uint32_t th = thr[i];
        const uint32_t fw = with_dp ? 7u : 4u;
        uint8_t *sb = text + off + plen;
        for (uint32_t s = threadIdx.x; s < S; s += blockDim.x) {
            const unsigned long long r0 = mix64(kv ^ ((unsigned long long)(2u * s) * 0x9E3779B97F4A7C15ull));
            const unsigned long long r1 = mix64(kv ^ ((unsigned long long)(2u * s + 1u) * 0x9E3779B97F4A7C15ull));
            const unsigned long long r2 = mix64(kv ^ ((unsigned long long)(2u * s) * 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A5A5A5A5A5ull);
            uint32_t a0 = (uint32_t)(r0 >> 32) < th ? (n_alt > 1u ? 1u + (uint32_t)((r0 >> 8) & 0xFFFFFFu) % n_alt : 1u) : 0u;
            uint32_t a1 = (uint32_t)(r1 >> 32) < th ? (n_alt > 1u ? 1u + (uint32_t)((r1 >> 8) & 0xFFFFFFu) % n_alt : 1u) : 0u;
            uint8_t c0 = (uint8_t)('0' + a0), c1 = (uint8_t)('0' + a1);
            const uint32_t mm = (uint32_t)(r2 & 0xFFFFu);
            if (mm < 1311u) c0 = c1 = '.';
            else if (mm < 1639u) {
                if ((r2 >> 16) & 1ull) c1 = '.';
                else c0 = '.';
            }
            const uint8_t sep = ((uint32_t)(r2 >> 20) & 0xFFFFu) < 3277u ? '/' : '|';
            uint8_t *f = sb + (unsigned long long)s * fw;
            f[0] = c0; f[1] = sep; f[2] = c1;
            uint32_t q = 3;
            if (with_dp) {
                const uint32_t dp = 10u + (uint32_t)((r2 >> 40) % 90ull);
                f[3] = ':'; f[4] = (uint8_t)('0' + dp / 10u); f[5] = (uint8_t)('0' + dp % 10u);
                q = 6;
            }
            f[q] = (s == S - 1u) ? '\n' : '\t';
        }
    }
}

extern "C" int hhgt_synth_render_mixed(hhgt_ctx *c, void *d_text, uint64_t text_cap, const uint64_t *d_line_off,
                                       const uint32_t *d_pos, const uint64_t *d_ref8, const uint64_t *d_alt8,
                                       const uint32_t *d_meta, const uint32_t *d_thr, uint64_t n_variants,
                                       uint64_t v_first, const char *contig, int n_samples, uint64_t seed, void *stream)
{
    if (!c || !d_text || !d_line_off || !d_pos || !d_ref8 || !d_alt8 || !d_meta || !d_thr || !contig || n_samples < 1) {
        hhgt_set_error("synth: bad arguments");
        return HHGT_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    SynthContig sc;
    memset(&sc, 0, sizeof(sc));
    size_t l = strlen(contig);
    if (l == 0 || l > 31) {
        hhgt_set_error("synth: contig name must be 1..31 bytes");
        return HHGT_ERR_ARG;
    }
    memcpy(sc.name, contig, l);
    sc.len = (int)l;
    if (n_variants == 0) return HHGT_OK;
    uint32_t grid = n_variants < 65536ull * 4 ? (uint32_t)n_variants : 65536u * 4u;
    hipLaunchKernelGGL(k_synth_mixed, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<uint8_t *>(d_text), text_cap, reinterpret_cast<const unsigned long long *>(d_line_off),
                       d_pos, reinterpret_cast<const unsigned long long *>(d_ref8),
                       reinterpret_cast<const unsigned long long *>(d_alt8), d_meta, d_thr, n_variants, v_first, sc,
                       (uint32_t)n_samples, mix64(seed + 0x9E3779B97F4A7C15ull));
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
