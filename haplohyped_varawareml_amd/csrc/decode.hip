// decode.hip — Blosc chunk decode (LZ4 block decode + byte-unshuffle) on gfx950.
//
// Read-side counterpart of lz4.hip/frame.hip: what the HDF5 filter 32001 does when the reference's
// reader pulls a dataset (/root/reference/src/utils/h5_reader.py:37-41) — and the device-side half of
// the encode -> compress -> decode round-trip property used by the full-size parity tests.
// Workgroup = one block; wave j decodes stream j in place: compressed bytes staged at the end of its LDS
// plane buffer, decoded bytes written from the start; the workgroup then un-shuffles the planes straight into HBM with 16 B stores.
#include "common.h"

#define BLOSC_DOSHUFFLE 0x1u
#define BLOSC_MEMCPYED 0x2u
#define BLOSC_DOBITSHUFFLE 0x4u
#define BLOSC_DONT_SPLIT 0x10u

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
// little-endian u32 at any byte address of GLOBAL memory (one unaligned dword load, not four byte loads)
__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { return *reinterpret_cast<const u32_unaligned *>(p); }

// a value every lane holds identically, moved to an SGPR: everything derived from it (ip, op, lengths, the
// branches on them) then compiles to scalar code instead of exec-masked vector code
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// LZ4 length extension starting at cin[ip]: returns added length, advances ip. wave-uniform.
__device__ __forceinline__ bool read_ext(const uint8_t *cin, uint32_t csize, uint32_t &ip, uint32_t &len)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        const uint32_t i = ip + lane;
        const uint32_t b = i < csize ? cin[i] : 0u;
        const unsigned long long stop = __ballot(b != 255u);
        if (stop == 0ull) {
            len += 255u * 64u;
            ip += 64u;
            if (ip >= csize) return false;
            continue;
        }
        const uint32_t j = (uint32_t)__builtin_ctzll(stop);
        if (ip + j >= csize) return false;
        len += 255u * j + (uint32_t)__builtin_amdgcn_readlane((int)b, (int)j);
        ip += j + 1u;
        return true;
    }
}

// out[op, op + ml) = the ml bytes starting `off` back (all arguments but lane wave-uniform, off >= 1)
__device__ __forceinline__ void lz4_match_copy(uint8_t *out, uint32_t op, uint32_t off, uint32_t ml, uint32_t lane)
{
    // byte k of the match is byte (k mod off) of the `off` bytes before op once the match overlaps itself, byte k of
    // them otherwise; one straight-line step serves the first 64 bytes of every case (a 64-byte step never reads a
    // byte written in the same step), the loop behind it is only entered by matches longer than that
    const uint8_t *pat = out + op - off;
    uint32_t ph = lane, step = 0u;
    if ((ml < 64u ? ml : 64u) > off) {  // off < 64 and ml > off: the match overlaps itself within a step
        if ((off & (off - 1u)) == 0u) {  // runs of a byte / pair / quad: the usual case, no division
            ph = lane & (off - 1u);
        } else {
            ph = lane % off;
            step = 64u % off;
        }
    }
    if (lane < ml) out[op + lane] = pat[ph];
    if (ml > 64u) {
        asm volatile("" ::: "memory");  // keep the (scalar) length test: the loop guard alone would cost every sequence
        if (off < 64u) {
            for (uint32_t k = lane + 64u; k < ml; k += 64u) {
                ph += step;
                if (ph >= off) ph -= off;
                out[op + k] = pat[ph];
            }
        } else {
            for (uint32_t k = lane + 64u; k < ml; k += 64u) out[op + k] = pat[k];
        }
    }
}

// decode cin[0, csize) (LDS) -> out[0, n) (LDS). returns true on success. wave-cooperative.
__device__ __forceinline__ bool lz4_wave_decode(const uint8_t *cin, uint32_t csize, uint8_t *out, uint32_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ip = 0, op = 0;
    for (;;) {
        // Fast path (nearly every sequence of a genotype plane): token, up to 13 literals, the offset and one match-length
        // extension byte all sit in the 19 bytes at ip.  Lane k (k <= 15) gathers the two aligned dwords around byte
        // ip + k and forms the 32-bit window starting there, so every header field is one readlane away: the token
        // from lane 0, offset + extension byte from lane ll + 1; byte 1 of lane k's window is literal k.  The kernel
        // is bound by scalar-ALU issue (the SQ counters show 1.0 scalar op per cycle per SIMD), so the point of this
        // shape is few scalar instructions per sequence.  (cin has >= 24 readable bytes past csize.)
        {
            const uint32_t q = ip + (lane < 15u ? lane : 15u);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(cin) + (q >> 2);
            const uint32_t v = __builtin_amdgcn_alignbyte(w[1], w[0], q & 3u);         // bytes ip + lane .. + 3
            const uint32_t t = uni(v);
            const uint32_t ll = (t >> 4) & 15u, mlt = t & 15u;
            if (ll <= 13u) {
                const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(ll + 1u));
                const uint32_t off = s3 & 0xFFFFu;
                // lengths 19..273 carry one extension byte (255 there = longer still: general path below); branch-free
                const uint32_t is15 = (mlt + 1u) >> 4;
                const uint32_t e1 = ((s3 >> 16) & 0xFFu) * is15;
                const uint32_t ml = mlt + 4u + e1, adv = ll + 3u + is15;
                // whole sequence inside the stream (so this is not the literal-only last one), output fits, offset
                // reaches no further back than the first byte; anything else is sorted out by the general path
                if (e1 != 255u && ip + adv <= csize && ll + ml <= n - op && off - 1u < op + ll) {
                    if (lane < ll) out[op + lane] = (uint8_t)(v >> 8);
                    op += ll;
                    lz4_match_copy(out, op, off, ml, lane);
                    op += ml;
                    ip += adv;
                    continue;
                }
            }
        }
        if (ip >= csize) return false;  // (the fast path cannot pass its own bounds check from here either)
        const uint32_t token = uni(cin[ip++]);
        uint32_t ll = token >> 4;
        if (ll == 15u && !read_ext(cin, csize, ip, ll)) return false;
        if (ll > csize - ip || ll > n - op) return false;
        for (uint32_t k = lane; k < ll; k += 64u) out[op + k] = cin[ip + k];
        op += ll;
        ip += ll;
        if (ip == csize) break;  // last sequence carries literals only
        if (csize - ip < 2u) return false;
        const uint32_t off = uni((uint32_t)cin[ip] | ((uint32_t)cin[ip + 1u] << 8));
        ip += 2u;
        if (off == 0u || off > op) return false;
        uint32_t ml = token & 15u;
        if (ml == 15u && !read_ext(cin, csize, ip, ml)) return false;
        ml += 4u;
        if (ml > n - op) return false;
        lz4_match_copy(out, op, off, ml, lane);
        op += ml;
    }
    return op == n;
}

// grid = n_chunks * nblocks; block = 64 * nwaves; dynamic LDS = nwaves * sstride + 16
__global__ __launch_bounds__(1024) void k_decode_blocks(const uint8_t *__restrict__ src,
                                                        const unsigned long long *__restrict__ chunk_off,
                                                        uint32_t nblocks, uint64_t chunk_nbytes, uint32_t typesize,
                                                        uint32_t blocksize, uint32_t sstride,
                                                        uint8_t *__restrict__ dst, unsigned long long *n_bad)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_bad;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;  // SGPR: stream sizes and LDS bases follow
    const uint64_t chunk = blockIdx.x / nblocks;
    const uint32_t b = blockIdx.x - (uint32_t)(chunk * nblocks);
    const uint8_t *ck = src + chunk_off[chunk];
    const uint32_t avail = (uint32_t)(chunk_off[chunk + 1] - chunk_off[chunk]);
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    bool bad = avail < 16u;
    uint32_t flags = 0, hl = 16, doshuffle = 0, dont_split = 1;
    if (!bad) {
        const uint32_t version = ck[0];
        flags = ck[2];
        const uint32_t ts = ck[3], nbytes = ld32u(ck + 4), bs = ld32u(ck + 8), cbytes = ld32u(ck + 12);
        const bool extended = (flags & BLOSC_DOSHUFFLE) && (flags & BLOSC_DOBITSHUFFLE);
        doshuffle = (flags & BLOSC_DOSHUFFLE) ? 1u : 0u;
        if (extended) {
            hl = 32;
            bad = bad || version < 3u || avail < 32u;
            doshuffle = 0;
            if (!bad)
                for (int k = 0; k < 6; ++k) {
                    uint32_t f = ck[16 + k];
                    if (f == 1u) doshuffle = 1u;
                    else if (f != 0u) bad = true;
                }
        } else if (flags & BLOSC_DOBITSHUFFLE)
            bad = true;
        dont_split = (flags & BLOSC_DONT_SPLIT) ? 1u : 0u;
        bad = bad || ts != typesize || nbytes != (uint32_t)chunk_nbytes || cbytes != avail;
        if (!(flags & BLOSC_MEMCPYED)) bad = bad || bs != blocksize || ((flags >> 5) & 7u) != 1u;
    }
    const uint64_t boff = (uint64_t)b * blocksize;
    const uint32_t bsize = (uint32_t)(chunk_nbytes - boff < blocksize ? chunk_nbytes - boff : blocksize);
    uint8_t *out_blk = dst + chunk * chunk_nbytes + boff;
    if (!bad && (flags & BLOSC_MEMCPYED)) {
        if (avail != chunk_nbytes + hl) bad = true;
        else
            for (uint32_t i = threadIdx.x; i < bsize; i += blockDim.x) out_blk[i] = ck[hl + boff + i];
        if (bad && threadIdx.x == 0 && b == 0) atomicAdd(n_bad, 1ull);
        return;
    }
    if (bad) {
        if (threadIdx.x == 0 && b == 0) atomicAdd(n_bad, 1ull);
        return;
    }
    const bool leftover = bsize != blocksize;
    const uint32_t nstreams = (!dont_split && !leftover) ? typesize : 1u;
    if (nstreams > nwaves) {  // header asks for a split this launch was not sized for
        if (threadIdx.x == 0 && b == 0) atomicAdd(n_bad, 1ull);
        return;
    }
    const uint32_t nelem = bsize / typesize;
    const uint32_t neblock = bsize / nstreams;
    const uint32_t pstride = nstreams > 1u ? sstride : nelem;
    uint8_t *planes = smem;
    // bytes of LDS the stream of this wave owns: its plane, or the whole area when the block is one stream
    const uint32_t cap = nstreams > 1u ? sstride : nwaves * sstride;
    // stream table of this block: walk the csize words (nstreams <= 16, serial by every thread)
    const uint32_t bstart = ld32u(ck + hl + 4u * b);
    uint32_t sp = bstart, my_sp = 0, my_cs = 0;
    bool sbad = false;
    for (uint32_t j = 0; j < nstreams; ++j) {
        if (sp + 4u > avail) {
            sbad = true;
            break;
        }
        const uint32_t cs = ld32u(ck + sp);
        if (cs == 0u || cs > neblock || sp + 4u + cs > avail) {
            sbad = true;
            break;
        }
        if (j == wave) {
            my_sp = sp + 4u;
            my_cs = cs;
        }
        sp += 4u + cs;
    }
    if (!sbad && wave < nstreams) {
        uint8_t *plane = planes + (size_t)wave * pstride;
        if (my_cs == neblock) {
            for (uint32_t k = lane; k < neblock; k += 64u) plane[k] = ck[my_sp + k];
        } else {
            // In-place decode (lz4.h's LZ4_DECOMPRESS_INPLACE_MARGIN scheme): the compressed bytes are staged at the
            // END of the plane's own buffer and the decoder writes from its start.  In a valid block the bytes still to
            // read never exceed the bytes still to write by more than 2 + n/255, so with a margin of (n >> 8) + 32 the
            // write head stays behind the read head; a malformed block can only garble its own output, every access
            // stays inside [plane, plane + cap).  Halves the LDS per workgroup -> 16 instead of 9 workgroups per CU.
            if (neblock + (neblock >> 8) + 60u > cap) sbad = true;
            uint8_t *cin = plane + ((cap - 24u - my_cs) & ~3u);   // dword-aligned; 24 readable bytes behind the stream
            const uint32_t nd = my_cs >> 2;       // whole dwords (unaligned global loads), then the last 1-3 bytes
            if (!sbad) {
                for (uint32_t k = lane; k < nd; k += 64u) reinterpret_cast<uint32_t *>(cin)[k] = ld32u(ck + my_sp + 4u * k);
                if (lane < (my_cs & 3u)) cin[4u * nd + lane] = ck[my_sp + 4u * nd + lane];
            }
            if (sbad || !lz4_wave_decode(cin, my_cs, plane, neblock)) sbad = true;
        }
    }
    if (sbad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (s_bad) {
        if (threadIdx.x == 0) atomicAdd(n_bad, 1ull);
        return;
    }
    // un-shuffle planes -> HBM
    if (!doshuffle || typesize == 1u) {
        for (uint32_t i = threadIdx.x; i < bsize; i += blockDim.x) out_blk[i] = planes[i];
    } else if (typesize == 2u && nstreams == 2u && (bsize & 15u) == 0u &&
               ((reinterpret_cast<uintptr_t>(out_blk) & 15u) == 0)) {
        for (uint32_t i = threadIdx.x * 16u; i < bsize; i += blockDim.x * 16u) {
            uint2 a = *reinterpret_cast<const uint2 *>(planes + (i >> 1));
            uint2 c = *reinterpret_cast<const uint2 *>(planes + pstride + (i >> 1));
            uint4 v;
            v.x = __builtin_amdgcn_perm(c.x, a.x, 0x05010400u);
            v.y = __builtin_amdgcn_perm(c.x, a.x, 0x07030602u);
            v.z = __builtin_amdgcn_perm(c.y, a.y, 0x05010400u);
            v.w = __builtin_amdgcn_perm(c.y, a.y, 0x07030602u);
            *reinterpret_cast<uint4 *>(out_blk + i) = v;
        }
    } else {
        const uint32_t body = nelem * typesize;
        for (uint32_t i = threadIdx.x; i < bsize; i += blockDim.x) {
            uint8_t v;
            if (i < body) {
                uint32_t e = i / typesize, j = i - e * typesize;
                v = planes[j * pstride + e];
            } else
                v = planes[(typesize - 1u) * pstride + nelem + (i - body)];
            out_blk[i] = v;
        }
    }
}

int launch_decode(const uint8_t *d_src, const uint64_t *d_chunk_off, uint64_t n_chunks, uint64_t chunk_nbytes,
                  int typesize, int blocksize, uint8_t *d_dst, unsigned long long *d_bad, hipStream_t st)
{
    if (n_chunks == 0) return HHGT_OK;
    const uint32_t split = (typesize >= 2 && typesize <= 16 && blocksize / typesize >= 128) ? 1u : 0u;
    const uint32_t nwaves = split ? (uint32_t)typesize : 1u;
    const uint32_t nblocks = (uint32_t)((chunk_nbytes + blocksize - 1) / blocksize);
    const uint32_t max_stream = split ? (uint32_t)blocksize / (uint32_t)typesize : (uint32_t)blocksize;
    // per-stream buffer = decoded plane + in-place margin ((n >> 8) + 32) + alignment slack (4) + read-ahead (24)
    uint32_t sstride = (max_stream + (max_stream >> 8) + 60u + 15u) & ~15u;
    {   // a block decoded as ONE stream (no-split header, or the short last block of a chunk) owns the whole area
        const uint32_t whole = (uint32_t)blocksize + ((uint32_t)blocksize >> 8) + 60u;
        if (whole > sstride * nwaves) sstride = ((whole + nwaves - 1) / nwaves + 15u) & ~15u;
    }
    const size_t lds = (size_t)nwaves * sstride + 16u;
    if (lds > 160 * 1024 - 64) {
        hhgt_set_error("decode: block of %d bytes x typesize %d does not fit LDS", blocksize, typesize);
        return HHGT_ERR_ARG;
    }
    static size_t attr_lds = 64 * 1024;  // dynamic LDS above 64 KiB needs an explicit opt-in
    if (lds > attr_lds) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_blocks),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const uint64_t grid = n_chunks * nblocks;
    if (grid > 0x7fffffffull) {
        hhgt_set_error("decode: too many blocks");
        return HHGT_ERR_ARG;
    }
    hipLaunchKernelGGL(k_decode_blocks, dim3((uint32_t)grid), dim3(64u * nwaves), lds, st, d_src,
                       reinterpret_cast<const unsigned long long *>(d_chunk_off), nblocks, chunk_nbytes,
                       (uint32_t)typesize, (uint32_t)blocksize, sstride, d_dst, d_bad);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
