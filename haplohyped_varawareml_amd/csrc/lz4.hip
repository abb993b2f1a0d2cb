// lz4.hip — Blosc byte-shuffle + LZ4 block encode, one Blosc block per workgroup (gfx950).
//
// Replaces the per-chunk work of the HDF5 filter the reference invokes at
//   create_dataset(..., compression=32001, compression_opts=(2,2,0,0,5,1,2))
//   /root/reference/src/haplohyped/vcf_to_h5.py:134-135
// i.e. c-blosc2's shuffle(typesize) -> split into `typesize` streams -> LZ4 block format per stream.
// The emitted bytes are a valid LZ4 block stream (not byte-identical to liblz4's: the contract is
// "decompresses to identical bytes").
//
// Workgroup = one block; wave j = stream j (byte plane j of the shuffled block when split).
//   phase A  all waves: coalesced 16 B/lane loads of the block, byte-plane de-interleave in registers
//            (v_perm_b32 for typesize 2), planes stored to LDS  -> shuffle costs no extra HBM traffic
//   phase B  wave j: greedy LZ4 over its plane in LDS.  64 positions are hashed/looked up/verified per
//            step (one position per lane), the first verified match is extended cooperatively
//            (64 bytes per ballot), sequences are emitted with lane-parallel byte stores.
// HBM roofline: algorithmic bytes = blocksize read + compressed bytes written (per block).
#include "common.h"

#define LZ_MINMATCH 4u
#define LZ_MFLIMIT 12u
#define LZ_LASTLITERALS 5u

size_t lz4_slot_bytes(int neblock)
{
    size_t b = (size_t)neblock + (size_t)neblock / 255 + 16;
    return (b + 15) & ~(size_t)15;
}

__device__ __forceinline__ uint32_t lds_load4(const uint8_t *in, uint32_t i)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(in);
    uint32_t a = w[i >> 2], b = w[(i >> 2) + 1];
    return __builtin_amdgcn_alignbyte(b, a, i & 3u);
}

__device__ __forceinline__ uint32_t emit_len(uint8_t *out, uint32_t q, uint32_t rem, uint32_t lane)
{
    // LZ4 length extension: rem/255 bytes of 255, then rem%255
    uint32_t nb = rem / 255u + 1u;
    for (uint32_t j = lane; j < nb; j += 64u) out[q + j] = (j == nb - 1u) ? (uint8_t)(rem % 255u) : (uint8_t)255u;
    return q + nb;
}

// Greedy LZ4 of in[0, n) (LDS, 4-byte aligned, >= 8 readable bytes past n) by one wave.
// Returns the compressed size (wave-uniform).  out has lz4_slot_bytes(n) capacity.
__device__ __forceinline__ uint32_t lz4_wave_compress(const uint8_t *in, uint32_t n, uint16_t *tab, uint32_t hashlog,
                                      uint8_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < (1u << hashlog) / 2u; i += 64u) reinterpret_cast<uint32_t *>(tab)[i] = 0u;
    uint32_t op = 0, anchor = 0;
    if (n > LZ_MFLIMIT) {
        const uint32_t mflimit = n - LZ_MFLIMIT, matchlimit = n - LZ_LASTLITERALS;
        const uint32_t hshift = 32u - hashlog;
        uint32_t p = 0;
        while (p <= mflimit) {
            const uint32_t i = p + lane;
            const bool valid = i <= mflimit;
            const uint32_t d = valid ? lds_load4(in, i) : 0u;
            const uint32_t h = (d * 2654435761u) >> hshift;
            uint32_t cand = valid ? (uint32_t)tab[h] : 0u;
            if (valid) tab[h] = (uint16_t)i;
            const uint32_t r = lds_load4(in, cand);
            const bool ok = valid && cand < i && r == d;
            unsigned long long m = __ballot(ok);
            const uint32_t wend = p + 64u;
            while (m) {
                const uint32_t sl = (uint32_t)__ffsll((long long)m) - 1u;
                uint32_t pos = p + sl;
                uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)sl);
                // forward extension, 64 bytes per step
                uint32_t ml = LZ_MINMATCH;
                for (;;) {
                    const uint32_t k = ml + lane;
                    const bool eq = (pos + k < matchlimit) && in[pos + k] == in[c + k];
                    const unsigned long long ne = __ballot(!eq);
                    if (ne == 0ull) {
                        ml += 64u;
                        continue;
                    }
                    ml += (uint32_t)__ffsll((long long)ne) - 1u;
                    break;
                }
                // backward extension over pending literals (at most 64 bytes)
                {
                    const uint32_t kk = lane + 1u;
                    const bool eq = (pos >= anchor + kk) && (c >= kk) && in[pos - kk] == in[c - kk];
                    const unsigned long long ne = __ballot(!eq);
                    const uint32_t nb = ne ? (uint32_t)__ffsll((long long)ne) - 1u : 64u;
                    pos -= nb;
                    c -= nb;
                    ml += nb;
                }
                // ---- emit sequence: token | literal-length ext | literals | offset | match-length ext
                const uint32_t ll = pos - anchor, mlc = ml - LZ_MINMATCH;
                if (lane == 0) out[op] = (uint8_t)(((ll < 15u ? ll : 15u) << 4) | (mlc < 15u ? mlc : 15u));
                uint32_t q = op + 1u;
                if (ll >= 15u) q = emit_len(out, q, ll - 15u, lane);
                for (uint32_t k = lane; k < ll; k += 64u) out[q + k] = in[anchor + k];
                q += ll;
                const uint32_t off = pos - c;
                if (lane == 0) out[q] = (uint8_t)(off & 0xFFu);
                if (lane == 1) out[q + 1u] = (uint8_t)(off >> 8);
                q += 2u;
                if (mlc >= 15u) q = emit_len(out, q, mlc - 15u, lane);
                op = q;
                anchor = pos + ml;
                if (anchor >= wend) m = 0ull;
                else m &= ~((1ull << (anchor - p)) - 1ull);
            }
            p = anchor > wend ? anchor : wend;
        }
    }
    // last literals
    {
        const uint32_t ll = n - anchor;
        if (lane == 0) out[op] = (uint8_t)((ll < 15u ? ll : 15u) << 4);
        uint32_t q = op + 1u;
        if (ll >= 15u) q = emit_len(out, q, ll - 15u, lane);
        for (uint32_t k = lane; k < ll; k += 64u) out[q + k] = in[anchor + k];
        op = q + ll;
    }
    return op;
}

// grid = n_chunks * nblocks; block = 64 * nwaves (nwaves = typesize when blocks are split, else 1)
// dynamic LDS: [data: nwaves * sstride + 16][tables: nwaves << (hashlog + 1)]
__global__ __launch_bounds__(1024) void k_lz4_blocks(const uint8_t *__restrict__ src, uint32_t nblocks,
                                                     uint64_t chunk_nbytes, uint32_t typesize, uint32_t blocksize,
                                                     uint32_t split, uint32_t sstride, uint32_t hashlog,
                                                     uint8_t *__restrict__ scratch, uint64_t slot_bytes,
                                                     uint32_t *__restrict__ csize)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t chunk = blockIdx.x / nblocks;
    const uint32_t b = blockIdx.x - (uint32_t)(chunk * nblocks);
    const uint64_t boff = (uint64_t)b * blocksize;
    const uint32_t bsize = (uint32_t)(chunk_nbytes - boff < blocksize ? chunk_nbytes - boff : blocksize);
    const bool leftover = bsize != blocksize;
    const uint32_t nstreams = (split && !leftover) ? typesize : 1u;
    const uint32_t nelem = bsize / typesize;
    const uint32_t neblock = bsize / nstreams;
    // plane stride in LDS: split streams are padded apart; a single stream keeps Blosc's contiguous
    // shuffled image [plane 0 | plane 1 | ... | tail]
    const uint32_t pstride = nstreams > 1u ? sstride : nelem;
    uint8_t *data = smem;
    uint16_t *tabs = reinterpret_cast<uint16_t *>(smem + (size_t)nwaves * sstride + 16u);
    const uint8_t *blk = src + chunk * chunk_nbytes + boff;

    // ---- phase A: load + byte-shuffle into LDS
    if (typesize == 1u) {
        for (uint32_t i = threadIdx.x * 16u; i < bsize; i += blockDim.x * 16u) {
            if (i + 16u <= bsize && ((reinterpret_cast<uintptr_t>(blk + i) & 15u) == 0))
                *reinterpret_cast<uint4 *>(data + i) = *reinterpret_cast<const uint4 *>(blk + i);
            else
                for (uint32_t j = i; j < bsize && j < i + 16u; ++j) data[j] = blk[j];
        }
    } else if (typesize == 2u && (bsize & 15u) == 0u && (pstride & 7u) == 0u &&
               ((reinterpret_cast<uintptr_t>(blk) & 15u) == 0)) {
        for (uint32_t i = threadIdx.x * 16u; i < bsize; i += blockDim.x * 16u) {
            uint4 v = *reinterpret_cast<const uint4 *>(blk + i);
            uint32_t p0a = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u), p0b = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u);
            uint32_t p1a = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u), p1b = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
            *reinterpret_cast<uint2 *>(data + (i >> 1)) = make_uint2(p0a, p0b);
            *reinterpret_cast<uint2 *>(data + pstride + (i >> 1)) = make_uint2(p1a, p1b);
        }
    } else {
        const uint32_t body = nelem * typesize;
        for (uint32_t i = threadIdx.x; i < bsize; i += blockDim.x) {
            uint8_t v = blk[i];
            if (i < body) {
                uint32_t e = i / typesize, j = i - e * typesize;
                data[j * pstride + e] = v;
            } else
                data[(typesize - 1u) * pstride + nelem + (i - body)] = v;  // tail after the last plane
        }
    }
    __syncthreads();
    // ---- phase B: one wave per stream
    if (wave < nstreams) {
        const uint8_t *in = data + (size_t)wave * pstride;
        const uint64_t sidx = (uint64_t)blockIdx.x * nwaves + wave;
        uint8_t *out = scratch + sidx * slot_bytes;
        uint32_t cs = lz4_wave_compress(in, neblock, tabs + ((size_t)wave << hashlog), hashlog, out);
        if (cs >= neblock) {  // incompressible: Blosc stores the (shuffled) stream verbatim
            for (uint32_t k = lane; k < neblock; k += 64u) out[k] = in[k];
            cs = neblock;
        }
        if (lane == 0) csize[sidx] = cs;
    } else if (lane == 0) {
        csize[(uint64_t)blockIdx.x * nwaves + wave] = 0u;
    }
}

int launch_lz4_blocks(const uint8_t *d_src, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize,
                      int blocksize, uint8_t *d_scratch, size_t slot_bytes, uint32_t *d_csize,
                      hipStream_t st)
{
    const uint32_t split = (typesize >= 2 && typesize <= 16 && blocksize / typesize >= 128) ? 1u : 0u;
    const uint32_t nwaves = split ? (uint32_t)typesize : 1u;
    const uint32_t nblocks = (uint32_t)((chunk_nbytes + blocksize - 1) / blocksize);
    // per-stream LDS stride: stream bytes + slack for the 4-byte lookahead, 16-byte aligned
    const uint32_t max_stream = split ? (uint32_t)blocksize / (uint32_t)typesize : (uint32_t)blocksize;
    uint32_t sstride = (max_stream + 8u + 15u) & ~15u;
    if (split && chunk_nbytes % blocksize) {
        // a leftover block is compressed as ONE stream laid out contiguously over the data area
        uint32_t need = ((uint32_t)(chunk_nbytes % blocksize) + 8u + 15u) & ~15u;
        if (need > sstride * nwaves) sstride = (need + nwaves - 1) / nwaves;
        sstride = (sstride + 15u) & ~15u;
    }
    const size_t data_bytes = (size_t)nwaves * sstride + 16u;
    uint32_t hashlog = 12;
    // keep >= 4 workgroups per CU resident when the block allows it (160 KiB LDS per CU)
    while (hashlog > 10 && data_bytes + ((size_t)nwaves << (hashlog + 1)) > 40960) --hashlog;
    const size_t lds = data_bytes + ((size_t)nwaves << (hashlog + 1));
    if (lds > 160 * 1024 - 64) {
        hhgt_set_error("lz4: block of %d bytes x typesize %d does not fit LDS", blocksize, typesize);
        return HHGT_ERR_ARG;
    }
    static size_t attr_lds = 64 * 1024;  // dynamic LDS above 64 KiB needs an explicit opt-in
    if (lds > attr_lds) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lz4_blocks),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const uint64_t grid = n_chunks * nblocks;
    if (grid == 0) return HHGT_OK;
    if (grid > 0x7fffffffull) {
        hhgt_set_error("lz4: too many blocks");
        return HHGT_ERR_ARG;
    }
    hipLaunchKernelGGL(k_lz4_blocks, dim3((uint32_t)grid), dim3(64u * nwaves), lds, st, d_src, nblocks,
                       chunk_nbytes, (uint32_t)typesize, (uint32_t)blocksize, split, sstride, hashlog, d_scratch,
                       (uint64_t)slot_bytes, d_csize);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
