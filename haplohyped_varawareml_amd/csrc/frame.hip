// frame.hip — Blosc chunk framing: header | bstarts | per block, per stream: int32 csize + bytes.
//
// Produces what c-blosc2 would hand back to the HDF5 filter pipeline for the call at
//   /root/reference/src/haplohyped/vcf_to_h5.py:134-135 (filter 32001, shuffle + LZ4-format codec):
// a self-describing chunk whose header says typesize / nbytes / blocksize / cbytes and whose
// bstarts[] give each block's offset, so a reader can decode one block (= one sample row of the
// genotype chunk) without touching the others.  Layout restated in oracle/codec_oracle.c, pinned
// there against c-blosc 1.21 for the 16-byte header form.
//
// k_frame_sizes : one workgroup per chunk; block sizes -> bstarts (exclusive scan), chunk cbytes,
//                 "memcpyed" decision (compressed form larger than nbytes + header).
// k_scan_u64    : chunk offsets (single workgroup; O(chunks)).
// k_frame_write : one workgroup per block; dword-wide funnel-shift copy of the stream bytes from the
//                 LZ4 scratch slots to their final (byte-aligned) position.
#include "common.h"

#define BLOSC_DOSHUFFLE 0x1u
#define BLOSC_MEMCPYED 0x2u
#define BLOSC_DOBITSHUFFLE 0x4u
#define BLOSC_DONT_SPLIT 0x10u

struct FrameParams {
    uint32_t nblocks, typesize, blocksize, split, nwaves, format, hl;
    uint64_t chunk_nbytes, slot_bytes;
};

__device__ __forceinline__ uint32_t frame_wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_frame_sizes(FrameParams P, const uint32_t *__restrict__ csize,
                                                     uint32_t *__restrict__ bstart,
                                                     unsigned long long *__restrict__ chunk_csize,
                                                     uint32_t *__restrict__ chunk_flags)
{
    __shared__ uint32_t sm[8];
    const uint64_t chunk = blockIdx.x;
    uint32_t carry = P.hl + 4u * P.nblocks;
    for (uint32_t b0 = 0; b0 < P.nblocks; b0 += 256u) {
        const uint32_t b = b0 + threadIdx.x;
        uint32_t sz = 0;
        if (b < P.nblocks) {
            const uint64_t boff = (uint64_t)b * P.blocksize;
            const bool leftover = P.chunk_nbytes - boff < P.blocksize;
            const uint32_t ns = (P.split && !leftover) ? P.typesize : 1u;
            const uint32_t *cs = csize + (chunk * P.nblocks + b) * P.nwaves;
            for (uint32_t j = 0; j < ns; ++j) sz += 4u + cs[j];
        }
        uint32_t inc = frame_wave_incl_scan(sz);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 63) sm[w] = inc;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t x = sm[i];
            if (i < w) base += x;
            tot += x;
        }
        __syncthreads();
        if (b < P.nblocks) bstart[chunk * P.nblocks + b] = carry + base + inc - sz;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        uint32_t flags = 0;
        unsigned long long cb = carry;
        if (cb > P.chunk_nbytes + P.hl) {
            flags = 1u;
            cb = P.chunk_nbytes + P.hl;
        }
        chunk_csize[chunk] = cb;
        chunk_flags[chunk] = flags;
    }
}

// single workgroup exclusive scan of uint64 (chunk sizes -> chunk offsets); out[n] = total
__global__ __launch_bounds__(256) void k_scan_u64(const unsigned long long *__restrict__ in, uint64_t n,
                                                  unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long sm[4];
    unsigned long long carry = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint64_t i0 = 0; i0 < n; i0 += 256u) {
        const uint64_t i = i0 + threadIdx.x;
        unsigned long long v = i < n ? in[i] : 0ull, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) sm[w] = inc;
        __syncthreads();
        unsigned long long base = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = sm[k];
            if (k < w) base += x;
            tot += x;
        }
        __syncthreads();
        if (i < n) out[i] = carry + base + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// copy n bytes src -> dst (dst arbitrary alignment, src 4-byte aligned) by one wave, in two phases so that a
// caller can issue the loads of SEVERAL streams before the first store: the kernel is a latency chain (one wave per
// block, a few hundred bytes per stream), so memory round trips per wave are what it costs.
struct WaveCopy {
    uint32_t a[4], b[4];  // first 1 KiB of the body: 4 dwords per lane
    uint32_t edge;        // head byte (lanes < head) or tail byte (lanes 32.. < 32 + tail)
    uint32_t head, nw, sh;
    bool small;
};

__device__ __forceinline__ void wave_copy_load(WaveCopy &c, const uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                               uint32_t n)
{
    const uint32_t tid = threadIdx.x & 63u;
    c.head = (uint32_t)((4u - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u);
    c.small = n < 16u + c.head;
    c.nw = c.small ? 0u : (n - c.head) >> 2;
    c.sh = c.head & 3u;  // src byte offset of dst word 0 (src is aligned, head < 4)
    c.edge = 0u;
    if (c.small) {
        if (tid < n) c.edge = src[tid];  // n < 19
        return;
    }
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    const uint32_t src_words = (n + 3u) >> 2;  // dwords of src that hold valid bytes
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = (uint32_t)k * 64u + tid;
        c.a[k] = w < c.nw ? s32[w] : 0u;
        c.b[k] = (c.sh && w < c.nw && w + 1u < src_words) ? s32[w + 1u] : 0u;
    }
    const uint32_t done = c.head + (c.nw << 2), tail = n - done;  // tail < 4
    if (tid < c.head) c.edge = src[tid];
    else if (tid >= 32u && tid - 32u < tail) c.edge = src[done + tid - 32u];
}

__device__ __forceinline__ void wave_copy_store(const WaveCopy &c, uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                uint32_t n)
{
    const uint32_t tid = threadIdx.x & 63u;
    if (c.small) {
        if (tid < n) dst[tid] = (uint8_t)c.edge;
        return;
    }
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + c.head);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = (uint32_t)k * 64u + tid;
        if (w < c.nw) d32[w] = __builtin_amdgcn_alignbyte(c.b[k], c.a[k], c.sh);
    }
    const uint32_t done = c.head + (c.nw << 2), tail = n - done;
    if (tid < c.head) dst[tid] = (uint8_t)c.edge;
    else if (tid >= 32u && tid - 32u < tail) dst[done + tid - 32u] = (uint8_t)c.edge;
    // streams longer than 1 KiB + head: the rest, 1 KiB per pass
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    const uint32_t src_words = (n + 3u) >> 2;
    for (uint32_t base = 256u; base < c.nw; base += 256u) {
        uint32_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = base + (uint32_t)k * 64u + tid;
            a[k] = w < c.nw ? s32[w] : 0u;
            b[k] = (c.sh && w < c.nw && w + 1u < src_words) ? s32[w + 1u] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = base + (uint32_t)k * 64u + tid;
            if (w < c.nw) d32[w] = __builtin_amdgcn_alignbyte(b[k], a[k], c.sh);
        }
    }
}

__device__ __forceinline__ void wg_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n)
{
    WaveCopy c;
    wave_copy_load(c, dst, src, n);
    wave_copy_store(c, dst, src, n);
}

__global__ __launch_bounds__(256) void k_frame_write(FrameParams P, const uint8_t *__restrict__ scratch,
                                                     const uint32_t *__restrict__ csize,
                                                     const uint8_t *__restrict__ src,
                                                     const uint32_t *__restrict__ bstart,
                                                     const unsigned long long *__restrict__ chunk_off,
                                                     const uint32_t *__restrict__ chunk_flags,
                                                     uint8_t *__restrict__ dst, uint64_t dst_cap,
                                                     uint64_t n_total_blocks, const uint8_t *__restrict__ planes, PlanesGeom pg)
{
    // one wave per Blosc block (blocks are a few KiB after compression: a whole workgroup per block idles)
    const uint64_t gb = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gb >= n_total_blocks) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t chunk = gb / P.nblocks;
    const uint32_t b = (uint32_t)(gb - chunk * P.nblocks);
    const unsigned long long coff = chunk_off[chunk], cend = chunk_off[chunk + 1];
    if (cend > dst_cap) return;  // capacity error is reported by the host from chunk_off[n]
    uint8_t *cdst = dst + coff;
    const uint32_t memcpyed = chunk_flags[chunk];
    const uint32_t cbytes = (uint32_t)(cend - coff);
    if (b == 0 && lane < P.hl) {
        // header bytes (see oracle/codec_oracle.c write_header)
        const uint32_t split_flag = P.split ? 0u : BLOSC_DONT_SPLIT;
        uint32_t flags = (1u << 5) | (P.typesize > 1u ? BLOSC_DOSHUFFLE : 0u) | split_flag | (memcpyed ? BLOSC_MEMCPYED : 0u);
        uint32_t i = lane;
        uint8_t v = 0;
        if (P.format == HHGT_BLOSC2) {
            if (i == 0) v = 5;
            else if (i == 1) v = 1;
            else if (i == 2) v = (uint8_t)(flags | BLOSC_DOSHUFFLE | BLOSC_DOBITSHUFFLE);
            else if (i == 3) v = (uint8_t)P.typesize;
            else if (i == 16 + 5) v = (flags & BLOSC_DOSHUFFLE) ? 1 : 0;
        } else {
            if (i == 0) v = 2;
            else if (i == 1) v = 1;
            else if (i == 2) v = (uint8_t)flags;
            else if (i == 3) v = (uint8_t)P.typesize;
        }
        if (i >= 4 && i < 8) v = (uint8_t)((uint32_t)P.chunk_nbytes >> ((i - 4) * 8));
        if (i >= 8 && i < 12) v = (uint8_t)(P.blocksize >> ((i - 8) * 8));
        if (i >= 12 && i < 16) v = (uint8_t)(cbytes >> ((i - 12) * 8));
        cdst[i] = v;
    }
    const uint64_t boff = (uint64_t)b * P.blocksize;
    const uint32_t bsize = (uint32_t)(P.chunk_nbytes - boff < P.blocksize ? P.chunk_nbytes - boff : P.blocksize);
    if (memcpyed) {
        // verbatim copy of the source block behind the header (source is 16-byte aligned per block
        // only when blocksize is; use the byte-safe path through wg_copy's 4-byte-aligned contract)
        const uint8_t *s = src + chunk * P.chunk_nbytes + boff;
        uint8_t *d = cdst + P.hl + boff;
        if (planes) {   // the block exists as bit planes (hhgt.h "Bit-plane form"): its bytes are generated (rare: an incompressible chunk)
            uint64_t pcol;
            uint32_t prow, pbi;
            planes_block(pg, gb, &pcol, &prow, &pbi);
            const uint64_t kst = (uint64_t)pg.S_pad * 32ull;   // bytes between the kind-planes of a tile
            for (uint32_t i = lane; i < 4096u; i += 64u) {
                const uint8_t *pl = planes + planes_piece(pg, pcol, pbi * 16u + (i >> 8), 0u, prow) + ((i & 255u) >> 3);
#pragma unroll
                for (uint32_t h = 0; h < 2u; ++h) {
                    uint32_t v = (pl[h * kst] >> (i & 7u)) & 1u;
                    if ((pl[(2ull + h) * kst] >> (i & 7u)) & 1u) v = v ? 0xF7u : (src ? (uint32_t)s[2u * i + h] : 0u);
                    d[2u * i + h] = (uint8_t)v;
                }
            }
        } else if ((reinterpret_cast<uintptr_t>(s) & 3u) == 0) wg_copy(d, s, bsize);
        else for (uint32_t i = lane; i < bsize; i += 64u) d[i] = s[i];
        return;
    }
    const uint32_t bs = bstart[chunk * P.nblocks + b];
    if (lane < 4) cdst[P.hl + 4u * b + lane] = (uint8_t)(bs >> (lane * 8));
    const bool leftover = bsize != P.blocksize;
    const uint32_t ns = (P.split && !leftover) ? P.typesize : 1u;
    uint32_t q = bs;
    // all stream sizes of the block in one load (lane j holds stream j; ns <= 16), broadcast per stream below
    const uint64_t sidx0 = ((uint64_t)chunk * P.nblocks + b) * P.nwaves;
    const uint32_t cs_l = lane < ns ? csize[sidx0 + lane] : 0u;
    if (ns == 2u) {  // the genotype case (typesize 2): both planes' loads in flight before the first store
        const uint32_t cs0 = (uint32_t)__builtin_amdgcn_readlane((int)cs_l, 0), cs1 = (uint32_t)__builtin_amdgcn_readlane((int)cs_l, 1);
        uint8_t *d0 = cdst + q + 4u, *d1 = d0 + cs0 + 4u;
        const uint8_t *s0 = scratch + sidx0 * P.slot_bytes, *s1 = s0 + P.slot_bytes;
        WaveCopy c0, c1;
        wave_copy_load(c0, d0, s0, cs0);
        wave_copy_load(c1, d1, s1, cs1);
        if (lane < 4) cdst[q + lane] = (uint8_t)(cs0 >> (lane * 8));
        else if (lane < 8) d0[cs0 + lane - 4u] = (uint8_t)(cs1 >> ((lane - 4u) * 8));
        wave_copy_store(c0, d0, s0, cs0);
        wave_copy_store(c1, d1, s1, cs1);
        return;
    }
    for (uint32_t j = 0; j < ns; ++j) {
        const uint64_t sidx = sidx0 + j;
        const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cs_l, (int)j);
        if (lane < 4) cdst[q + lane] = (uint8_t)(cs >> (lane * 8));
        wg_copy(cdst + q + 4u, scratch + sidx * P.slot_bytes, cs);
        q += 4u + cs;
    }
}

int launch_frame(const uint8_t *d_scratch, size_t slot_bytes, const uint32_t *d_csize, const uint8_t *d_src, const uint8_t *d_planes,
                 PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize, int format,
                 uint32_t *d_bstart, uint64_t *d_chunk_csize, uint8_t *d_dst, uint64_t dst_cap,
                 uint64_t *d_chunk_off, uint32_t *d_chunk_flags, hipStream_t st)
{
    if (n_chunks == 0) return HHGT_OK;
    FrameParams P;
    P.nblocks = (uint32_t)((chunk_nbytes + blocksize - 1) / blocksize);
    P.typesize = (uint32_t)typesize;
    P.blocksize = (uint32_t)blocksize;
    P.split = (typesize >= 2 && typesize <= 16 && blocksize / typesize >= 128) ? 1u : 0u;
    P.nwaves = P.split ? (uint32_t)typesize : 1u;
    P.format = (uint32_t)format;
    P.hl = format == HHGT_BLOSC2 ? 32u : 16u;
    P.chunk_nbytes = chunk_nbytes;
    P.slot_bytes = slot_bytes;
    hipLaunchKernelGGL(k_frame_sizes, dim3((uint32_t)n_chunks), dim3(256), 0, st, P, d_csize, d_bstart,
                       reinterpret_cast<unsigned long long *>(d_chunk_csize), d_chunk_flags);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(256), 0, st,
                       reinterpret_cast<const unsigned long long *>(d_chunk_csize), n_chunks,
                       reinterpret_cast<unsigned long long *>(d_chunk_off));
    const uint64_t n_total_blocks = n_chunks * P.nblocks;
    hipLaunchKernelGGL(k_frame_write, dim3((uint32_t)((n_total_blocks + 3) / 4)), dim3(256), 0, st, P, d_scratch,
                       d_csize, d_src, d_bstart, reinterpret_cast<const unsigned long long *>(d_chunk_off),
                       d_chunk_flags, d_dst, dst_cap, n_total_blocks, d_planes, pg);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
