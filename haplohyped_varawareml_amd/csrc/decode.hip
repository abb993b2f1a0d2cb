// decode.hip — Blosc chunk decode (LZ4 block decode + byte-unshuffle) on gfx950.
//
// Read-side counterpart of lz4.hip/frame.hip: what the HDF5 filter 32001 does when the reference's
// reader pulls a dataset (/root/reference/src/utils/h5_reader.py:37-41) — and the device-side half of
// the encode -> compress -> decode round-trip property used by the full-size parity tests.
// Workgroup = one block; wave j decodes stream j from a staged LDS copy of its compressed bytes into
// its LDS plane; the workgroup then un-shuffles the planes straight into HBM with 16 B stores.
#include "common.h"

#define BLOSC_DOSHUFFLE 0x1u
#define BLOSC_MEMCPYED 0x2u
#define BLOSC_DOBITSHUFFLE 0x4u
#define BLOSC_DONT_SPLIT 0x10u

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

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
        const uint32_t j = (uint32_t)__ffsll((long long)stop) - 1u;
        if (ip + j >= csize) return false;
        len += 255u * j + (uint32_t)__builtin_amdgcn_readlane((int)b, (int)j);
        ip += j + 1u;
        return true;
    }
}

// decode cin[0, csize) (LDS) -> out[0, n) (LDS). returns true on success. wave-cooperative.
__device__ __forceinline__ bool lz4_wave_decode(const uint8_t *cin, uint32_t csize, uint8_t *out, uint32_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ip = 0, op = 0;
    for (;;) {
        if (ip >= csize) return false;
        const uint32_t token = cin[ip++];
        uint32_t ll = token >> 4;
        if (ll == 15u && !read_ext(cin, csize, ip, ll)) return false;
        if (ll > csize - ip || ll > n - op) return false;
        for (uint32_t k = lane; k < ll; k += 64u) out[op + k] = cin[ip + k];
        op += ll;
        ip += ll;
        if (ip == csize) break;  // last sequence carries literals only
        if (csize - ip < 2u) return false;
        const uint32_t off = (uint32_t)cin[ip] | ((uint32_t)cin[ip + 1u] << 8);
        ip += 2u;
        if (off == 0u || off > op) return false;
        uint32_t ml = token & 15u;
        if (ml == 15u && !read_ext(cin, csize, ip, ml)) return false;
        ml += 4u;
        if (ml > n - op) return false;
        if (off >= 64u) {
            // 64-byte steps never read bytes written in the same step
            for (uint32_t k = lane; k < ml; k += 64u) out[op + k] = out[op + k - off];
        } else {
            // overlapping match = periodic extension of the last `off` bytes
            const uint8_t *pat = out + op - off;
            uint32_t ph = lane % off;
            const uint32_t step = 64u % off;
            for (uint32_t k = lane; k < ml; k += 64u) {
                out[op + k] = pat[ph];
                ph += step;
                if (ph >= off) ph -= off;
            }
        }
        op += ml;
    }
    return op == n;
}

// grid = n_chunks * nblocks; block = 64 * nwaves; dynamic LDS = nwaves * (pstride + cstride) + 32
__global__ __launch_bounds__(1024) void k_decode_blocks(const uint8_t *__restrict__ src,
                                                        const unsigned long long *__restrict__ chunk_off,
                                                        uint32_t nblocks, uint64_t chunk_nbytes, uint32_t typesize,
                                                        uint32_t blocksize, uint32_t sstride, uint32_t cstride,
                                                        uint8_t *__restrict__ dst, unsigned long long *n_bad)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_bad;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
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
    uint8_t *cbuf = smem + (size_t)nwaves * sstride + 16u;
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
            uint8_t *cin = cbuf + (size_t)wave * cstride;
            for (uint32_t k = lane; k < my_cs; k += 64u) cin[k] = ck[my_sp + k];
            if (!lz4_wave_decode(cin, my_cs, plane, neblock)) sbad = true;
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
    uint32_t sstride = (max_stream + 8u + 15u) & ~15u;
    if (split && chunk_nbytes % blocksize) {
        uint32_t need = ((uint32_t)(chunk_nbytes % blocksize) + 8u + 15u) & ~15u;
        if (need > sstride * nwaves) sstride = (need + nwaves - 1) / nwaves;
        sstride = (sstride + 15u) & ~15u;
    }
    // compressed staging: a stream never exceeds its decoded size; the leftover/no-split stream of a
    // split launch is staged across the whole staging area
    uint32_t cstride = sstride;
    const size_t lds = (size_t)nwaves * sstride + 16u + (size_t)nwaves * cstride + 16u;
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
                       (uint32_t)typesize, (uint32_t)blocksize, sstride, cstride, d_dst, d_bad);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
