// scan.hip — device-wide exclusive prefix sum of uint32 (reduce / scan-partials / downsweep).
// Small helper used by the line index, the kept-record compaction and the chunk framing; the
// arrays are O(lines) or O(chunks), never O(text bytes), so this is not a bandwidth-critical kernel.
#include "common.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix,
// *total = block sum (valid in all threads)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *sm /*>=8*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; ++i) {
        uint32_t x = sm[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce(const uint32_t *__restrict__ in, uint64_t n,
                                                            uint32_t *__restrict__ partial)
{
    HHGT_WAVE_PRIO();
    __shared__ uint32_t sm[8];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint64_t idx = base + (uint64_t)i * SCAN_BLOCK + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    uint32_t tot;
    block_excl_scan(s, &tot, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single block: exclusive scan of partial[0..nb) in place, partial[nb] = grand total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(uint32_t *partial, uint32_t nb)
{
    HHGT_WAVE_PRIO();
    __shared__ uint32_t sm[8];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
        uint32_t i = b0 + threadIdx.x;
        uint32_t v = i < nb ? partial[i] : 0;
        uint32_t tot;
        uint32_t ex = block_excl_scan(v, &tot, sm);
        if (i < nb) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) partial[nb] = carry;
}

// out[i] = partial[block] + exclusive prefix inside the tile; out[n] = grand total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_down(const uint32_t *__restrict__ in, uint64_t n,
                                                          const uint32_t *__restrict__ partial, uint32_t nb,
                                                          uint32_t *__restrict__ out)
{
    HHGT_WAVE_PRIO();
    __shared__ uint32_t sm[8];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint64_t idx = base + i;
        v[i] = idx < n ? in[idx] : 0;
        s += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan(s, &tot, sm) + partial[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint64_t idx = base + i;
        if (idx < n) out[idx] = ex;
        ex += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[nb];
}

// (declared in common.h)
size_t scan_tmp_elems(uint64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 2; }

// d_out must hold n + 1 elements (d_out[n] = total).  In-place (d_in == d_out) is NOT supported.
int launch_scan_exclusive_u32(const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp,
                              size_t tmp_elems, hipStream_t st)
{
    uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb == 0) nb = 1;
    if (tmp_elems < (size_t)nb + 1) {
        hhgt_set_error("scan: tmp too small");
        return HHGT_ERR_ARG;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, st, d_in, n, d_tmp);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(SCAN_BLOCK), 0, st, d_tmp, nb);
    hipLaunchKernelGGL(k_scan_down, dim3(nb), dim3(SCAN_BLOCK), 0, st, d_in, n, d_tmp, nb, d_out);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// ---- two launches instead of three, and two arrays per pass: the arrays scanned on the encode path have at most a few
// hundred tiles, so every downsweep block sums the partials in front of it itself (no k_scan_partials in between), and
// the keep / chrom-run flags of the lines go through together.  B = 1 or 2 arrays.
template <int B>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce_n(const uint32_t *__restrict__ in0, const uint32_t *__restrict__ in1,
                                                              uint64_t n, uint32_t *__restrict__ partial, uint32_t nb)
{
    HHGT_WAVE_PRIO();
    __shared__ uint32_t sm[8];
    const uint32_t *ins[2] = {in0, in1};
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const uint64_t idx = base + (uint64_t)i * SCAN_BLOCK + threadIdx.x;
            if (idx < n) s += ins[b][idx];
        }
        uint32_t tot;
        block_excl_scan(s, &tot, sm);
        if (threadIdx.x == 0) partial[(size_t)b * nb + blockIdx.x] = tot;
    }
}

template <int B>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_down_n(const uint32_t *__restrict__ in0, const uint32_t *__restrict__ in1,
                                                            uint64_t n, const uint32_t *__restrict__ partial, uint32_t nb,
                                                            uint32_t *__restrict__ out0, uint32_t *__restrict__ out1)
{
    HHGT_WAVE_PRIO();
    __shared__ uint32_t sm[8];
    const uint32_t *ins[2] = {in0, in1};
    uint32_t *outs[2] = {out0, out1};
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        // sum of the tiles in front of this one
        uint32_t pre = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += SCAN_BLOCK) pre += partial[(size_t)b * nb + i];
        uint32_t front;
        block_excl_scan(pre, &front, sm);
        uint32_t v[SCAN_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const uint64_t idx = base + i;
            v[i] = idx < n ? ins[b][idx] : 0;
            s += v[i];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan(s, &tot, sm) + front;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            const uint64_t idx = base + i;
            if (idx < n) outs[b][idx] = ex;
            ex += v[i];
        }
        if (blockIdx.x == nb - 1 && threadIdx.x == 0) outs[b][n] = front + tot;
    }
}

// out_a[0..n] / out_b[0..n] = exclusive prefix sums of in_a / in_b (out[n] = total); in_b == nullptr: one array.
// d_tmp: 2 * scan_tmp_elems(n) elements.  More than SCAN_SHORT_MAX tiles: the three-launch form, one array at a time.
#define SCAN_SHORT_MAX 2048u
int launch_scan_exclusive_u32_pair(const uint32_t *in_a, uint32_t *out_a, const uint32_t *in_b, uint32_t *out_b, uint64_t n,
                                   uint32_t *d_tmp, size_t tmp_elems, hipStream_t st)
{
    uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb == 0) nb = 1;
    if (nb > SCAN_SHORT_MAX) {
        int rc = launch_scan_exclusive_u32(in_a, out_a, n, d_tmp, tmp_elems, st);
        if (rc == HHGT_OK && in_b) rc = launch_scan_exclusive_u32(in_b, out_b, n, d_tmp, tmp_elems, st);
        return rc;
    }
    if (tmp_elems < 2 * (size_t)nb) {
        hhgt_set_error("scan: tmp too small");
        return HHGT_ERR_ARG;
    }
    if (in_b) {
        hipLaunchKernelGGL(k_scan_reduce_n<2>, dim3(nb), dim3(SCAN_BLOCK), 0, st, in_a, in_b, n, d_tmp, nb);
        hipLaunchKernelGGL(k_scan_down_n<2>, dim3(nb), dim3(SCAN_BLOCK), 0, st, in_a, in_b, n, d_tmp, nb, out_a, out_b);
    } else {
        hipLaunchKernelGGL(k_scan_reduce_n<1>, dim3(nb), dim3(SCAN_BLOCK), 0, st, in_a, in_a, n, d_tmp, nb);
        hipLaunchKernelGGL(k_scan_down_n<1>, dim3(nb), dim3(SCAN_BLOCK), 0, st, in_a, in_a, n, d_tmp, nb, out_a, out_a);
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

