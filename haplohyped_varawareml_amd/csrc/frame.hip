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
#include <stdlib.h>

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

// copy n bytes src -> dst (dst arbitrary alignment, src 16-byte aligned: an LZ4 scratch slot) by one wave, in two phases so that
// a caller can issue the loads of SEVERAL streams before the first store: the kernel is a latency chain (one wave per block, a
// few hundred bytes per stream), so memory round trips and instructions per wave are what it costs.  Round 4: 16 bytes per
// lane — one load instruction per KiB — stored where they belong with byte-aligned 16-byte stores (the first version moved
// dwords, two loads per dword and a funnel shift to reach dword-aligned destinations: 8 load instructions per KiB).
typedef uint32_t u32x4_fr __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_fr_unaligned __attribute__((ext_vector_type(4), aligned(1)));
struct WaveCopy {
    u32x4_fr a;   // bytes [16 lane, 16 lane + 16) of the first KiB
};

__device__ __forceinline__ void wave_copy_load(WaveCopy &c, const uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                               uint32_t n)
{
    const uint32_t o = (threadIdx.x & 63u) * 16u;
    c.a = u32x4_fr{0u, 0u, 0u, 0u};
    // (loading the slot's first KiB whatever n is — the load then waits for no size, one round trip less per wave — was
    // measured: 1.94 against 1.74 ms per step; the kernel is bound by what it moves, not by its chain)
    if (o < n) c.a = *reinterpret_cast<const u32x4_fr *>(src + o);   // (a slot is padded to 16 bytes: the last piece may read past n)
}

__device__ __forceinline__ void wave_put16(uint8_t *__restrict__ d, const u32x4_fr &v, uint32_t left)
{
    if (left >= 16u) {
        *reinterpret_cast<u32x4_fr_unaligned *>(d) = v;
        return;
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t k = 0; k < 15u; ++k)
        if (k < left) d[k] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
}

__device__ __forceinline__ void wave_copy_store(const WaveCopy &c, uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                uint32_t n)
{
    const uint32_t o = (threadIdx.x & 63u) * 16u;
    if (o < n) wave_put16(dst + o, c.a, n - o);
    // streams longer than 1 KiB: the rest, 4 KiB per pass with the loads in flight together
    for (uint32_t base = 1024u; base < n; base += 4096u) {
        u32x4_fr v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = base + (uint32_t)k * 1024u + o;
            v[k] = u32x4_fr{0u, 0u, 0u, 0u};
            if (q < n) v[k] = *reinterpret_cast<const u32x4_fr *>(src + q);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = base + (uint32_t)k * 1024u + o;
            if (q < n) wave_put16(dst + q, v[k], n - q);
        }
    }
}

__device__ __forceinline__ void wg_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n)
{
    WaveCopy c;
    wave_copy_load(c, dst, src, n);
    wave_copy_store(c, dst, src, n);
}

// header (block 0), bstarts entry and streams of Blosc block b of chunk `chunk`, by ONE wave.  coff / cend: where the chunk
// starts and ends in dst; bs: the block's offset inside the chunk; gb: the block's index in the call.
__device__ __forceinline__ void frame_block(const FrameParams &P, const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ csize,
                                            const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t chunk, uint32_t b,
                                            uint64_t gb, unsigned long long coff, unsigned long long cend, uint32_t memcpyed, uint32_t bs,
                                            const uint8_t *__restrict__ planes, const PlanesGeom &pg)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *cdst = dst + coff;
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
        } else if ((reinterpret_cast<uintptr_t>(s) & 15u) == 0 && (bsize & 15u) == 0) wg_copy(d, s, bsize);
        else for (uint32_t i = lane; i < bsize; i += 64u) d[i] = s[i];
        return;
    }
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
        // (the stores below start at any byte; a timing-only build that rounded d0 / d1 down to 16 bytes ran the stage in the
        // same 1.73-1.75 ms: the alignment of the stores is not what the kernel waits for, so a destination-driven gather
        // with aligned stores was not built)
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

__global__ __launch_bounds__(256) void k_frame_write(FrameParams P, const uint8_t *__restrict__ scratch,
                                                     const uint32_t *__restrict__ csize,
                                                     const uint8_t *__restrict__ src,
                                                     const uint32_t *__restrict__ bstart,
                                                     const unsigned long long *__restrict__ chunk_off,
                                                     const uint32_t *__restrict__ chunk_flags,
                                                     uint8_t *__restrict__ dst, uint64_t dst_cap,
                                                     uint64_t n_total_blocks, const uint8_t *__restrict__ planes, PlanesGeom pg)
{
    // one wave per Blosc block (blocks are a few KiB after compression: a whole workgroup per block idles).  (Round 4: a
    // wave per PAIR of blocks, the four streams in flight together, was slower — 2.05 against 1.76 ms per 3 M x 2504 step
    // alone, 5.5 against 2.8 ms beside the encode chain: the kernel lives on the number of waves in flight.)
    const uint64_t gb = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gb >= n_total_blocks) return;
    const uint64_t chunk = gb / P.nblocks;
    const uint32_t b = (uint32_t)(gb - chunk * P.nblocks);
    const unsigned long long coff = chunk_off[chunk], cend = chunk_off[chunk + 1];
    if (cend > dst_cap) return;  // capacity error is reported by the host from chunk_off[n]
    const uint32_t memcpyed = chunk_flags[chunk];
    frame_block(P, scratch, csize, src, dst, chunk, b, gb, coff, cend, memcpyed, memcpyed ? 0u : bstart[chunk * P.nblocks + b], planes, pg);
}

// ---------------------------------------------------------------------------------------------
// k_frame_fused (round 4): the three kernels above in ONE launch.  FR_SPLIT workgroups per chunk; each sums the sizes of
// its chunk's blocks (bstarts: an exclusive scan over at most FR_MAXB blocks, kept in LDS), the first of them chains the
// chunk totals into chunk offsets by a decoupled look-back over one 64-bit state word per chunk (status | launch tag |
// value: the word carries its own data, so no fence is needed beside agent-scope atomic loads / stores), the others wait
// for that word; then every wave copies its share of the blocks to their final place.  Chunk ids are handed out by a
// ticket in start order, so a workgroup only ever waits for workgroups that already run.  The LZ4 kernels are complete
// when this starts (stream order): nobody waits for data, only for the offsets of at most a few chunks in front.
#define FR_MAXB 1024u
#define FR_ST_AGG 1ull
#define FR_ST_INC 2ull
#define FR_VAL_BITS 42
#define FR_TAG_BITS 20
struct FrameState {
    unsigned long long ticket;
    unsigned long long pad_[7];
    unsigned long long st[1];   // [n_chunks]
};

__device__ __forceinline__ unsigned long long fr_word(unsigned long long status, uint32_t tag, unsigned long long v)
{
    return (status << 62) | ((unsigned long long)tag << FR_VAL_BITS) | v;
}

__global__ __launch_bounds__(256) void k_frame_fused(FrameParams P, const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ csize,
                                                     const uint8_t *__restrict__ src, FrameState *__restrict__ fs, uint32_t tag,
                                                     uint32_t fr_split, unsigned long long *__restrict__ chunk_off, uint8_t *__restrict__ dst,
                                                     uint64_t dst_cap, uint64_t n_chunks, const uint8_t *__restrict__ planes, PlanesGeom pg)
{
    __shared__ uint32_t s_bstart[FR_MAXB];
    __shared__ uint32_t sm[8];
    __shared__ unsigned long long s_tk, s_coff;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(&fs->ticket, 1ull);
        if (t + 1ull == n_chunks * fr_split) fs->ticket = 0ull;   // the last one to start: every other ticket is drawn, the next launch starts at 0
        s_tk = t;
    }
    __syncthreads();
    const uint64_t chunk = s_tk / fr_split;
    const uint32_t q = (uint32_t)(s_tk - chunk * fr_split);
    // ---- block sizes of the chunk -> bstarts, chunk size (every workgroup of the chunk: 256 loads)
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
        const uint32_t inc = frame_wave_incl_scan(sz);
        if (lane == 63) sm[w] = inc;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = sm[i];
            if ((uint32_t)i < w) base += x;
            tot += x;
        }
        __syncthreads();
        if (b < P.nblocks) s_bstart[b] = carry + base + inc - sz;
        carry += tot;
    }
    unsigned long long cb = carry;
    const uint32_t memcpyed = cb > P.chunk_nbytes + P.hl ? 1u : 0u;
    if (memcpyed) cb = P.chunk_nbytes + P.hl;
    // ---- chunk offset
    unsigned long long *st = fs->st;
    if (w == 0) {
        unsigned long long coff = 0;
        if (q == 0) {
            if (chunk != 0) {
                if (lane == 0) __hip_atomic_store(&st[chunk], fr_word(FR_ST_AGG, tag, cb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // look back: 64 chunks per step, the nearest inclusive word ends it
                long long hi = (long long)chunk - 1;
                for (;;) {
                    const long long i = hi - (long long)lane;
                    unsigned long long x = 0;
                    bool ok = false;
                    for (;;) {   // until every word down to the nearest inclusive one (or 64 of them) is of this launch
                        x = i >= 0 ? __hip_atomic_load(&st[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : fr_word(FR_ST_INC, tag, 0ull);
                        const uint32_t xt = (uint32_t)(x >> FR_VAL_BITS) & ((1u << FR_TAG_BITS) - 1u);
                        ok = xt == tag && (x >> 62) != 0ull;
                        const unsigned long long incm = __builtin_amdgcn_ballot_w64(ok && (x >> 62) == FR_ST_INC);
                        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
                        // lanes in front of (= nearer than) the first inclusive one must all be there
                        const unsigned long long need = incm ? ((incm & (0ull - incm)) << 1) - 1ull : ~0ull;
                        if ((okm & need) == need) {
                            const bool mine = ((need >> lane) & 1ull) != 0ull;
                            unsigned long long v = mine ? (x & ((1ull << FR_VAL_BITS) - 1ull)) : 0ull;
#pragma unroll
                            for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
                            coff += (unsigned long long)__shfl(v, 0, 64);
                            ok = incm != 0ull;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (ok) break;
                    hi -= 64;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(&st[chunk], fr_word(FR_ST_INC, tag, coff + cb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                chunk_off[chunk] = coff;
                if (chunk + 1 == n_chunks) chunk_off[n_chunks] = coff + cb;
            }
        } else {
            for (;;) {
                const unsigned long long x = __hip_atomic_load(&st[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t xt = (uint32_t)(x >> FR_VAL_BITS) & ((1u << FR_TAG_BITS) - 1u);
                if (xt == tag && (x >> 62) == FR_ST_INC) {
                    coff = (x & ((1ull << FR_VAL_BITS) - 1ull)) - cb;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (lane == 0) s_coff = coff;
    }
    __syncthreads();
    const unsigned long long coff = s_coff, cend = coff + cb;
    if (cend > dst_cap) return;   // capacity error is reported by the host from chunk_off[n]
    // ---- this workgroup's share of the blocks, a wave per block
    const uint32_t per = (P.nblocks + fr_split - 1u) / fr_split;
    const uint32_t b_end = (q + 1u) * per < P.nblocks ? (q + 1u) * per : P.nblocks;
    for (uint32_t b = q * per + w; b < b_end; b += 4u)
        frame_block(P, scratch, csize, src, dst, chunk, b, chunk * P.nblocks + b, coff, cend, memcpyed, memcpyed ? 0u : s_bstart[b], planes, pg);
}

int launch_frame(const uint8_t *d_scratch, size_t slot_bytes, const uint32_t *d_csize, const uint8_t *d_src, const uint8_t *d_planes,
                 PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize, int format,
                 uint32_t *d_bstart, uint64_t *d_chunk_csize, uint8_t *d_dst, uint64_t dst_cap,
                 uint64_t *d_chunk_off, uint32_t *d_chunk_flags, void *d_state, uint32_t tag, hipStream_t st)
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
    // HHGT_FRAME_FUSED=1: the one-launch form (k_frame_fused) — same bytes (tests/test_gpu_frame_fused.py), built and measured
    // in round 4: framing 2.97 against 1.83 ms per 3 M x 2504 step (a ticket per workgroup on one word, a look-back chain and
    // 8 blocks per wave against one) — so the three launches stay the default
    static const bool fused_env = getenv("HHGT_FRAME_FUSED") && atoi(getenv("HHGT_FRAME_FUSED")) != 0;
    if (fused_env && d_state && P.nblocks <= FR_MAXB && n_chunks < (1ull << 30) &&
        n_chunks * (chunk_nbytes + 32) < (1ull << FR_VAL_BITS)) {
        // workgroups per chunk: about 8 blocks per wave
        uint32_t split = (P.nblocks + 31u) / 32u;
        split = split < 1u ? 1u : (split > 8u ? 8u : split);
        hipLaunchKernelGGL(k_frame_fused, dim3((uint32_t)(n_chunks * split)), dim3(256), 0, st, P, d_scratch, d_csize, d_src,
                           static_cast<FrameState *>(d_state), tag, split, reinterpret_cast<unsigned long long *>(d_chunk_off), d_dst, dst_cap,
                           n_chunks, d_planes, pg);
        HIP_TRY(hipGetLastError());
        return HHGT_OK;
    }
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

size_t frame_state_bytes(uint64_t n_chunks) { return sizeof(FrameState) + (size_t)n_chunks * 8u; }
