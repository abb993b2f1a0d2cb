// onehot.hip — haplotype windows -> one-hot float32 tensors on the device (BASELINE config 5).
//
// Replaces RandomHaplotypeDataset.encode_haplotypes + encode_sequence
//   /root/reference/src/datasets/haplotype_dataset.py:86-110   (allele==1 -> ALT else REF at pos-start)
//   /root/reference/src/utils/common_utils.py:84-103           (upper-case, non-ACGT -> N, one-hot by key order)
// k_overlay : one thread per (item, variant in window): writes the haplotype's base at the variant's
//             window offset into a sparse byte overlay (0 = "no variant here").
// k_onehot  : flat float4 stores over [items][seq_len * C] — fully coalesced 16 B/lane writes; the kernel
//             is HBM-write bound: algorithmic bytes = 2 * items * seq_len * C * 4 written
//             (+ items * seq_len * 3 read).
#include "common.h"

struct OneHotLut {
    uint8_t v[256];
};

__global__ __launch_bounds__(256) void k_overlay(const hhgt_window *__restrict__ items, uint32_t seq_len,
                                                 uint8_t *__restrict__ ovl1, uint8_t *__restrict__ ovl2)
{
    const uint32_t b = blockIdx.y;
    const hhgt_window w = items[b];
    const uint32_t *vs = reinterpret_cast<const uint32_t *>(w.var_start_ptr);
    const uint8_t *vr = reinterpret_cast<const uint8_t *>(w.var_ref_ptr);
    const uint8_t *va = reinterpret_cast<const uint8_t *>(w.var_alt_ptr);
    const int8_t *g = reinterpret_cast<const int8_t *>(w.geno_ptr);
    for (uint32_t j = w.var_lo + blockIdx.x * blockDim.x + threadIdx.x; j < w.var_hi; j += gridDim.x * blockDim.x) {
        const long long off = (long long)vs[j] - w.win_start;
        if (off < 0 || off >= (long long)seq_len) continue;
        // several records at one position: the last one wins (np.put_along_axis order), deterministically
        if (j + 1 < w.var_hi && vs[j + 1] == vs[j]) continue;
        const int8_t h0 = g[2ull * (j - w.geno_first)], h1 = g[2ull * (j - w.geno_first) + 1];
        const uint8_t r = vr[j], a = va[j];
        ovl1[(size_t)b * seq_len + off] = h0 == 1 ? a : r;   // haplotype_dataset.py:99
        ovl2[(size_t)b * seq_len + off] = h1 == 1 ? a : r;   // haplotype_dataset.py:100
    }
}

__device__ __forceinline__ uint32_t base_channel(const hhgt_window &w, const uint8_t *ovl, size_t oi, uint32_t i,
                                                 const OneHotLut &lut)
{
    uint8_t o = ovl[oi];
    if (o == 0) {
        const unsigned long long gp = (unsigned long long)(w.win_start + (long long)i);
        o = (w.win_start + (long long)i >= 0 && gp < w.ref_len) ? reinterpret_cast<const uint8_t *>(w.ref_ptr)[gp]
                                                                 : (uint8_t)'N';
    }
    return lut.v[o];
}

// one thread = 4 consecutive floats of the flattened [seq_len * C] row of one item, both haplotypes
__global__ __launch_bounds__(256) void k_onehot(const hhgt_window *__restrict__ items, uint32_t seq_len, uint32_t C,
                                                OneHotLut lut, const uint8_t *__restrict__ ovl1,
                                                const uint8_t *__restrict__ ovl2, float *__restrict__ hap1,
                                                float *__restrict__ hap2)
{
    const uint32_t b = blockIdx.y;
    const hhgt_window w = items[b];
    const uint64_t n = (uint64_t)seq_len * C;
    const uint64_t nq = (n + 3) / 4;
    float *o1 = hap1 + (size_t)b * n, *o2 = hap2 + (size_t)b * n;
    const bool vec_ok = (n & 3ull) == 0ull;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e0 = 4 * q;
        float f1[4], f2[4];
        uint32_t i = (uint32_t)(e0 / C), c = (uint32_t)(e0 - (uint64_t)i * C);
        uint32_t ch1 = 256, ch2 = 256, cur_i = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e0 + k < n) {
                if (i != cur_i) {
                    ch1 = base_channel(w, ovl1, (size_t)b * seq_len + i, i, lut);
                    ch2 = base_channel(w, ovl2, (size_t)b * seq_len + i, i, lut);
                    cur_i = i;
                }
                f1[k] = c == ch1 ? 1.0f : 0.0f;
                f2[k] = c == ch2 ? 1.0f : 0.0f;
            } else {
                f1[k] = f2[k] = 0.0f;
            }
            if (++c == C) {
                c = 0;
                ++i;
            }
        }
        if (vec_ok) {
            *reinterpret_cast<float4 *>(o1 + e0) = make_float4(f1[0], f1[1], f1[2], f1[3]);
            *reinterpret_cast<float4 *>(o2 + e0) = make_float4(f2[0], f2[1], f2[2], f2[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e0 + k < n) {
                    o1[e0 + k] = f1[k];
                    o2[e0 + k] = f2[k];
                }
        }
    }
}

extern "C" int hhgt_onehot_windows(hhgt_ctx *c, const hhgt_window *d_items, uint32_t n_items, uint32_t seq_len,
                                   const uint8_t *lut, int n_channels, float *d_hap1, float *d_hap2, void *stream)
{
    if (!c || !d_items || !lut || !d_hap1 || !d_hap2 || n_channels < 1 || n_channels > 254 || seq_len == 0) {
        hhgt_set_error("onehot: bad arguments");
        return HHGT_ERR_ARG;
    }
    if (n_items == 0) return HHGT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    const size_t ovl_bytes = (size_t)n_items * seq_len;
    int rc = c->oh_ovl.ensure(2 * ovl_bytes);
    if (rc != HHGT_OK) return rc;
    OneHotLut L;
    for (int i = 0; i < 256; ++i) L.v[i] = lut[i] < (uint8_t)n_channels ? lut[i] : (uint8_t)255;
    uint8_t *ovl1 = c->oh_ovl.as<uint8_t>(), *ovl2 = ovl1 + ovl_bytes;
    StageTimer t(c, st, HHGT_STAGE_ONEHOT);
    HIP_TRY(hipMemsetAsync(ovl1, 0, 2 * ovl_bytes, st));
    hipLaunchKernelGGL(k_overlay, dim3(8, n_items), dim3(256), 0, st, d_items, seq_len, ovl1, ovl2);
    const uint64_t nq = ((uint64_t)seq_len * n_channels + 3) / 4;
    uint32_t gx = (uint32_t)((nq + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_onehot, dim3(gx, n_items), dim3(256), 0, st, d_items, seq_len, (uint32_t)n_channels, L, ovl1,
                       ovl2, d_hap1, d_hap2);
    t.stop();
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// ---------------------------------------------------------------------------------------------
// bases -> uint8 one-hot rows [n][C] (reference genome store).  One thread = 4 output bytes.
__global__ __launch_bounds__(256) void k_onehot_bases_u8(const uint8_t *__restrict__ bases, uint64_t n, uint32_t C,
                                                         OneHotLut lut, uint8_t *__restrict__ out)
{
    const uint64_t total = n * C;
    const uint64_t nq = (total + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e0 = 4 * q;
        uint64_t i = e0 / C;
        uint32_t c = (uint32_t)(e0 - i * C);
        uint32_t ch = 256, word = 0;
        uint64_t cur_i = ~0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e0 + k < total) {
                if (i != cur_i) {
                    ch = lut.v[bases[i]];
                    cur_i = i;
                }
                word |= (c == ch ? 1u : 0u) << (8 * k);
            }
            if (++c == C) {
                c = 0;
                ++i;
            }
        }
        if (e0 + 4 <= total) *reinterpret_cast<uint32_t *>(out + e0) = word;
        else
            for (int k = 0; k < 4; ++k)
                if (e0 + k < total) out[e0 + k] = (uint8_t)(word >> (8 * k));
    }
}

extern "C" int hhgt_onehot_bases_u8(hhgt_ctx *c, const uint8_t *d_bases, uint64_t n, const uint8_t *lut, int n_channels,
                                    uint8_t *d_out, void *stream)
{
    if (!c || !d_bases || !lut || !d_out || n_channels < 1 || n_channels > 254) {
        hhgt_set_error("onehot_bases: bad arguments");
        return HHGT_ERR_ARG;
    }
    if (n == 0) return HHGT_OK;
    if (reinterpret_cast<uintptr_t>(d_out) & 3u) {
        hhgt_set_error("onehot_bases: d_out must be 4-byte aligned");
        return HHGT_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    OneHotLut L;
    for (int i = 0; i < 256; ++i) L.v[i] = lut[i] < (uint8_t)n_channels ? lut[i] : (uint8_t)255;
    const uint64_t nq = (n * (uint64_t)n_channels + 3) / 4;
    uint32_t gx = (uint32_t)((nq + 255) / 256 < 65536 ? (nq + 255) / 256 : 65536);
    StageTimer t(c, reinterpret_cast<hipStream_t>(stream), HHGT_STAGE_ONEHOT);
    hipLaunchKernelGGL(k_onehot_bases_u8, dim3(gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_bases, n,
                       (uint32_t)n_channels, L, d_out);
    t.stop();
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
