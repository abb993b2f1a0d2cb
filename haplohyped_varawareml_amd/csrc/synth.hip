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
