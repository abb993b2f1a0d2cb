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
//   phase B  wave j: greedy LZ4 over its plane in LDS, 64 positions (one per lane) per window
//            (lz4_wave_compress_v6): every lane verifies and measures its own candidates — the hash table's
//            most recent occurrence and the offset-1 run; the greedy left-to-right choice of non-overlapping
//            matches is the only serial part and runs on the scalar unit; the chosen sequences are queued in LDS
//            and laid out (DPP prefix sum) and written one per lane, 48-64 at a time.
//            lz4_wave_compress (v1: one match per round trip) is kept as a second instantiation for A/B runs
//            (HHGT_LZ4_ALGO=1).
// Algorithmic bytes = blocksize read + compressed bytes written per block; measured: the kernel is bound by
// instruction issue, not by HBM.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

#define LZ_MINMATCH 4u
#define LZ_MFLIMIT 12u
#define LZ_LASTLITERALS 5u
// Shortest offset-1 run / hash match the window encoder takes as a sequence.  LZ4 allows 4; on the genotype workload
// a 4-5 byte match saves 1-2 bytes but ends the literal run and often pre-empts a better match one position later
// (the parse is greedy): measured ratio 4.326 with 4/4, 4.399 with 5/5, **4.410 with 6/6**, 4.352 with 7/7,
// 4.231 with 8/8 — and fewer sequences are less work (35.4 -> 35.0 ms).  Override with -DLZ_MINRUN= / -DLZ_MINHASH=.
#ifndef LZ_BACK_STEPS
#define LZ_BACK_STEPS 1   // backward-extension rounds of the high-effort mode
#endif
#ifndef LZ_MINRUN
#define LZ_MINRUN 6u
#endif
// Bytes behind the end of every stream in LDS that are ZERO: the window encoder's lanes fetch the 24 bytes at positions
// up to mflimit + 63 = n + 51, i.e. they look at up to 76 bytes past the end of their stream.  Those positions are never
// coded, but what a lane in front of them measures (how far its hash candidate agrees, whether the offset-1 run or the
// hash match is the longer one before both are cut at the stream's match limit) decided between two equally valid
// encodings — and with 8 bytes of slack the second wave of a workgroup was looking at the first wave's live hash table
// there: streams differed from run to run on a few planes (round 3, tools/dev/planes_diff.py).  Fixed in round 4.
#define LZ_SLACK 96u
#ifndef LZ_MINHASH
#define LZ_MINHASH 6u
#endif

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

// ---------------------------------------------------------------------------------------------
// helpers of the window-parallel encoder (its history, v2 .. v6: DESIGN.md §3.1).
// A lane fetches the 24 bytes at its own position and the 24 bytes at its candidate as aligned dwords
// (ds_read2_b32 x 3 per side) and measures its match up to 20 bytes from registers.
struct Own6 {
    uint32_t w[6];
};

__device__ __forceinline__ Own6 lds_load6(const uint8_t *in, uint32_t pos)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (pos >> 2);
    Own6 o;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.w[k] = w[k];
    return o;
}

__device__ __forceinline__ uint32_t div255(uint32_t x) { return (x * 0x8081u) >> 23; }  // exact for x < 65536

// ---------------------------------------------------------------------------------------------
// DPP row shifts and the wave-wide inclusive scan built from them (used by the flush)
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t x, int d)
{
    switch (d) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    }
}

// inclusive prefix sum over the 64 lanes; *total = wave sum (uniform)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x, uint32_t lane, uint32_t *total)
{
    x += dpp_row_shr(x, 1);
    x += dpp_row_shr(x, 2);
    x += dpp_row_shr(x, 4);
    x += dpp_row_shr(x, 8);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31) + t0;
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + t1;
    const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 63) + t2;
    const uint32_t row = lane >> 4;
    x += row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t1 : t2));
    *total = t3;
    return x;
}

#ifdef HHGT_LZ4_STATS
// development build only (tools/lz4_stats.py): event counts of the window loop
__device__ unsigned long long g_lz4_stat[8];
#define LZ_STAT(i, v)                                                                \
    do {                                                                             \
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_lz4_stat[i], (unsigned long long)(v)); \
    } while (0)
extern "C" int hhgt_debug_lz4_stats(unsigned long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lz4_stat), sizeof(g_lz4_stat)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lz4_stat), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define LZ_STAT(i, v) ((void)0)
#endif

// v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0 (kept opaque so that min() folds stay one instruction)
__device__ __forceinline__ uint32_t ffbl(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// number of equal leading bytes (0..4) given x = a ^ b
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x)
{
    const uint32_t t = ffbl(x) >> 3;  // x == 0 -> 0x1FFFFFFF
    return t < 4u ? t : 4u;
}
// NOTE on unaligned LDS reads: gfx950 accepts ds_read_b32/b64/b128 at any byte address, which would remove every
// v_alignbyte below — but tools/micro/lds_unaligned.hip measures ~57 cycles per unaligned wave-instruction
// (lanes serialised, any width) against 3-5 aligned; the encoder built that way ran 3.3x slower.  Aligned
// dwords + v_alignbyte it is.
struct __attribute__((packed)) PackedU32 {
    uint32_t v;
};
__device__ __forceinline__ unsigned long long ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// Layout + emission of the queued sequences, one per lane (lane j = j-th match in stream order).
// queue[j] = (pos | len << 16, offset).  Updates op (output size so far) and anchor (end of the last match).
__device__ __forceinline__ void lz4_flush_queue(const uint8_t *in, uint8_t *__restrict__ out, const uint2 *queue,
                                                uint32_t qn, uint32_t &op, uint32_t &anchor, uint32_t lane)
{
    const bool valid = lane < qn;
    const uint2 e = queue[lane];  // stale entries beyond qn are masked below
    const uint32_t pos = e.x & 0xFFFFu, len = e.x >> 16, off = e.y;
    const uint32_t endp = pos + len;
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)endp, 0x138, 0xf, 0xf, false);  // wave_shr:1
    const uint32_t prev_end = lane ? up : anchor;
    const uint32_t ll = valid ? pos - prev_end : 0u, mlc = len - LZ_MINMATCH;
    const uint32_t llx = ll >= 15u ? div255(ll - 15u) + 1u : 0u;
    const uint32_t mlx = (valid && mlc >= 15u) ? div255(mlc - 15u) + 1u : 0u;
    const uint32_t size = valid ? 3u + llx + ll + mlx : 0u;
    uint32_t total;
    const uint32_t so = op + wave_incl_scan_dpp(size, lane, &total) - size;
    if (valid) {
        const uint32_t token = ((ll < 15u ? ll : 15u) << 4) | (mlc < 15u ? mlc : 15u);
        const uint32_t q = so + 1u + llx + ll;
        out[so] = (uint8_t)token;
        out[so + (llx == 1u ? 1u : 0u)] = (uint8_t)(llx == 1u ? ll - 15u : token);
        out[q] = (uint8_t)(off & 0xFFu);
        out[q + 1u] = (uint8_t)(off >> 8);
        out[q + (mlx == 1u ? 2u : 1u)] = (uint8_t)(mlx == 1u ? mlc - 15u : off >> 8);
    }
    if (ballot((llx | mlx) > 1u) != 0ull) {  // rare: 255-runs
        if (valid && (llx | mlx) > 1u) {
            const uint32_t q = so + 1u + llx + ll;
            if (llx > 1u) {
                const uint32_t rem = ll - 15u;
                for (uint32_t k = 0; k < llx; ++k) out[so + 1u + k] = (k == llx - 1u) ? (uint8_t)(rem - 255u * (llx - 1u)) : (uint8_t)255u;
            }
            if (mlx > 1u) {
                const uint32_t rem = mlc - 15u;
                for (uint32_t k = 0; k < mlx; ++k) out[q + 2u + k] = (k == mlx - 1u) ? (uint8_t)(rem - 255u * (mlx - 1u)) : (uint8_t)255u;
            }
        }
    }
    // literals: short runs by their own lane, long runs (> 32 bytes: incompressible stretches) by the whole wave
    const uint32_t lit_dst = so + 1u + llx;
    unsigned long long big = ballot(ll > 32u);
    while (big) {
        const uint32_t j = (uint32_t)__ffsll((long long)big) - 1u;
        big &= big - 1ull;
        const uint32_t n_l = (uint32_t)__builtin_amdgcn_readlane((int)ll, (int)j);
        const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)prev_end, (int)j);
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)lit_dst, (int)j);
        for (uint32_t k = lane; k < n_l; k += 64u) out[dst + k] = in[src + k];
    }
    // short runs: 4 bytes per step from two aligned LDS dwords; whole dwords go out as one (unaligned) global
    // store, the last 1-3 bytes one by one (the bytes after them belong to other lanes)
    const uint32_t lshort = ll > 32u ? 0u : ll;
    const uint32_t lsh = prev_end & 3u;
    for (uint32_t k = 0; ballot(k < lshort) != 0ull; k += 4u) {
        if (k < lshort) {
            const uint32_t *wsrc = reinterpret_cast<const uint32_t *>(in) + ((prev_end + k) >> 2);
            const uint32_t v = __builtin_amdgcn_alignbyte(wsrc[1], wsrc[0], lsh);
            const uint32_t rem = lshort - k;
            uint8_t *dstp = out + lit_dst + k;
            if (rem >= 4u) {
                reinterpret_cast<PackedU32 *>(dstp)->v = v;
            } else {
                dstp[0] = (uint8_t)v;
                if (rem > 1u) dstp[1] = (uint8_t)(v >> 8);
                if (rem > 2u) dstp[2] = (uint8_t)(v >> 16);
            }
        }
    }
    op += total;
    anchor = (uint32_t)__builtin_amdgcn_readlane((int)endp, (int)(qn - 1u));
}

// v6: v4's search and parse (history: DESIGN.md §3.1) with deferred, batched emission (lz4_flush_queue).
// FAST = run candidate only (no hash table, no LDS gathers): the low-clevel mode, like LZ4's acceleration.
//
// PMC of this kernel (tools/pmc_lz4.sh): ~6.7 k VALU + ~5.3 k SALU wave-instructions per 4 KiB stream and the
// SIMDs' VALU issue slots > 85 % busy — the vector ALU is the bound, LDS (0.6 k) and the scalar unit have
// room.  So per-lane predicates live as wave masks in SGPR pairs (v_cmp writes them for free), everything
// wave-uniform (range limits, the position-0 exclusion, end-of-stream clipping) is scalar and branched
// around, and the enqueue runs under exec = SEL set by two scalar moves instead of a per-lane test.
// hash of the NB bytes at a lane's position (own = its six aligned dwords, sh = pos & 3, d = the first four bytes)
template <int NB>
__device__ __forceinline__ uint32_t lz4_key_hash(const Own6 &own, uint32_t sh, uint32_t d)
{
    if constexpr (NB <= 4) return d * 2654435761u;
    uint32_t kx = d;
    constexpr int rot[4] = {13, 7, 21, 27};
#pragma unroll
    for (int k = 1; k < (NB + 3) / 4; ++k) {
        uint32_t e = __builtin_amdgcn_alignbyte(own.w[k + 1], own.w[k], sh);
        if (k == (NB + 3) / 4 - 1 && (NB & 3)) e &= (1u << (8 * (NB & 3))) - 1u;   // key ends inside this dword
        kx ^= __builtin_amdgcn_alignbit(e, e, rot[k - 1]);
    }
    return kx * 2246822519u;
}

// Key lengths.  The default mode's single table is keyed on 12 bytes, not LZ4's 4: with a minimum match of 6 the short
// key only buys candidates that die early, and the most recent place where the next TWELVE bytes were the same is
// far more often the start of a long match (genotype planes: ratio 4.41 -> 5.12 and 2 % FASTER; 6 / 8 / 10 / 16 / 20 key
// bytes: 4.60 / 4.85 / 5.01 / 4.77 / 4.54).  A second table with another key length on top (tried: 4..12 with 12..20)
// added 1 % for 8 ms and was dropped.
#ifndef LZ_KEY
#define LZ_KEY 12
#endif

template <bool FAST, bool LONGRUN>
__device__ __forceinline__ uint32_t lz4_wave_compress_v6(const uint8_t *in, uint32_t n, uint16_t *tab,
                                                         uint32_t hashlog, uint8_t *__restrict__ out, uint2 *queue)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < (1u << hashlog) / 2u + 1u; i += 64u) reinterpret_cast<uint32_t *>(tab)[i] = 0u;
    const uint32_t to_end = 64u - lane;
    const uint32_t queue_lds = lds_addr(queue);
    uint32_t op = 0, anchor = 0, qn = 0;
    if (n > LZ_MFLIMIT) {
        const uint32_t mflimit = n - LZ_MFLIMIT, matchlimit = n - LZ_LASTLITERALS;
        const uint32_t hshift = 32u - hashlog;
        const uint32_t lim63 = mflimit - 63u;  // p - 1 >= lim63 (unsigned) <=> p == 0 or p + 63 > mflimit
        uint32_t p = 0;
        Own6 own = lds_load6(in, lane);
        // byte before this lane's position (0x100 = "none": position 0 continues no run)
        uint32_t pbv = lane ? (uint32_t)in[lane - 1u] : 0x100u;
        uint32_t best_start = 0u, best_len = 0u, best_byte = 0x100u;  // longest extended run so far (none yet)
        LZ_STAT(0, 1);
        while (p <= mflimit) {
            LZ_STAT(1, 1);
            const uint32_t pos = p + lane;
            const uint32_t sh = pos & 3u;
            const uint32_t d = __builtin_amdgcn_alignbyte(own.w[1], own.w[0], sh);
            uint32_t hcand = 0u, hsh = 0u;
            Own6 cw;
            if constexpr (!FAST) {
                // positions past mflimit are inserted too: only the last window has them and nothing reads
                // the table after it
                const uint32_t h = lz4_key_hash<LZ_KEY>(own, sh, d) >> hshift;
                hcand = (uint32_t)tab[h];
                tab[h] = (uint16_t)pos;
                cw = lds_load6(in, hcand);
                hsh = hcand & 3u;
            }
            // two candidates per position: (a) the hash table's most recent occurrence of these 4 bytes,
            // (b) offset 1 — the position continues a run of one byte value.  (b) needs no table and no LDS:
            // E = ballot(byte[pos] == byte[pos-1]) is one compare per window, and a lane's run length is the
            // number of consecutive set bits of E from its own bit (exact up to the window end).  On sparse
            // genotype planes (b) roughly halves what a zero run costs, because an earlier "0000" copy breaks
            // wherever the EARLIER text had a 1 (ratio 3.36 -> 4.35 on the 3 M x 2504 workload).
            const unsigned long long E = ballot((d & 0xFFu) == pbv);
            const unsigned long long r = ~E >> lane;  // first set bit = end of this lane's run (none: the window end)
            const uint32_t f_lo = ffbl((uint32_t)r), f_hi = ffbl((uint32_t)(r >> 32)) | 32u;
            const uint32_t f = f_lo < f_hi ? f_lo : f_hi;
            const uint32_t run = f < to_end ? f : to_end;
            // per-lane predicates are wave masks (SGPR pairs written by v_cmp), combined on the scalar unit
            const unsigned long long Rm = ballot(run >= LZ_MINRUN);
            unsigned long long Hm = 0ull;
            if constexpr (!FAST) Hm = ballot(d == __builtin_amdgcn_alignbyte(cw.w[1], cw.w[0], hsh));
            // the table starts zeroed, so hcand < pos except at position 0; candidates need no range test:
            // position 0 and lanes past mflimit are taken out of M on the scalar side, in the two windows that have them
            unsigned long long M = Hm | Rm, range_m = ~0ull;
            if (p - 1u >= lim63) {
                asm volatile("" ::: "memory");  // keep this a (rarely taken) scalar branch
                if (p == 0u) range_m &= ~1ull;
                if (p + 63u > mflimit) range_m &= ~0ull >> (63u - (mflimit - p));
                M &= range_m;
            }
            if (M == 0ull) {
                LZ_STAT(2, 1);
                p += 64u;
                own = lds_load6(in, p + lane);
                pbv = (uint32_t)in[p + lane - 1u];
                continue;
            }
            uint32_t lenh = 0u;
            unsigned long long Zm = 0ull;
            if constexpr (!FAST) {
                const uint32_t x1 = __builtin_amdgcn_alignbyte(own.w[2], own.w[1], sh) ^ __builtin_amdgcn_alignbyte(cw.w[2], cw.w[1], hsh);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(own.w[3], own.w[2], sh) ^ __builtin_amdgcn_alignbyte(cw.w[3], cw.w[2], hsh);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(own.w[4], own.w[3], sh) ^ __builtin_amdgcn_alignbyte(cw.w[4], cw.w[3], hsh);
                const uint32_t x4 = __builtin_amdgcn_alignbyte(own.w[5], own.w[4], sh) ^ __builtin_amdgcn_alignbyte(cw.w[5], cw.w[4], hsh);
                // equal bytes of x1..x4: the first non-zero dword decides (all zero: 16 + 4 = 20, still matching)
                const uint32_t t1 = x1 ? x1 : (x2 ? x2 : (x3 ? x3 : x4));
                const uint32_t skip = x1 ? 4u : (x2 ? 8u : (x3 ? 12u : 16u));
                const uint32_t lraw = skip + eq_bytes(t1);
                lenh = __builtin_amdgcn_inverse_ballot_w64(Hm) ? lraw : 0u;
                Zm = ballot((x1 | x2 | x3 | x4) == 0u);
            }
            const uint32_t lenr = __builtin_amdgcn_inverse_ballot_w64(Rm) ? run : 0u;
            if constexpr (!FAST) {
                // A hash match that starts 1 or 2 bytes before a run and ends inside it is a bad buy: "match, then
                // the rest of the run" is two sequences (>= 6 bytes) where "1-2 literals, then the whole run" is one
                // (<= 5).  On sparse genotype planes that is every '1' followed by zeros (the typical window):
                // such candidates are dropped, and with them the reason to measure hash matches past 20 bytes
                // (a second compare batch for bytes 20..35 ran in 80 % of the windows, tools/lz4_stats.py).
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)lenr, 0x130, 0xf, 0xf, true);  // wave_shl:1, 0 shifted in
                const uint32_t r2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r1, 0x130, 0xf, 0xf, true);
                // Third candidate: the first byte of a run (it differs from its predecessor, so offset 1 cannot cover
                // it) copied together with the run from the LONGEST EARLIER RUN of the same byte value.  The hash
                // table's most recent "0000" sits at the end of the previous run, four bytes before the '1' that ends
                // it — a source that dies after 4 bytes; LZ4HC's chain search finds the long one, which is why c-blosc's
                // lz4hc (what the reference's compression_opts select) packs these planes far tighter than lz4.  The
                // wave remembers the longest extended run so far (start, length, byte value; scalar registers).
                // With the variant below: +26 % kernel time for +7.6 % ratio on the bench workload — the higher-effort
                // mode (clevel >= 7), not the default.
                const unsigned long long Lm0 = LONGRUN ? ballot((d & 0xFFu) == best_byte) & ~E & (Rm >> 1) & range_m : 0ull;
                if (LONGRUN && Lm0 != 0ull) {
                    const uint32_t want = r1 + 1u;                                    // this byte + the run behind it
                    const uint32_t lenl = want < best_len ? want : best_len;
                    const unsigned long long Lm = Lm0 & ballot(lenl > lenh);
                    const bool isl = __builtin_amdgcn_inverse_ballot_w64(Lm);
                    lenh = isl ? lenl : lenh;
                    hcand = isl ? best_start : hcand;
                    Hm |= Lm;
                    M |= Lm;
                    // still matching: the run reaches the window end and the source run is longer than that
                    Zm = (Zm & ~Lm) | (Lm & ballot(want == to_end) & ballot(best_len > want));
                }
                // ... and every candidate is tried one byte to the left as its left neighbour's candidate (LZ4HC's backward
                // extension, lane-parallel): if the byte in front of lane+1's source equals this lane's byte, this lane
                // has a match one byte longer from one byte earlier.  That turns "1, then 0 0 0 ... from the long run"
                // into one match with no literal, and lengthens ordinary hash matches found one position late.
                if constexpr (LONGRUN)
#pragma unroll
                for (int step = 0; step < LZ_BACK_STEPS; ++step) {
                    const uint32_t nh = (uint32_t)__builtin_amdgcn_mov_dpp((int)hcand, 0x130, 0xf, 0xf, true);   // lane+1's source
                    const uint32_t nl = (uint32_t)__builtin_amdgcn_mov_dpp((int)lenh, 0x130, 0xf, 0xf, true);    // ... and length
                    const uint32_t pbyte = (uint32_t)in[nh ? nh - 1u : 0u];
                    const unsigned long long Bm = ballot(nl + 1u > lenh) & ballot(nh != 0u) & ballot((d & 0xFFu) == pbyte) & (Hm >> 1) & range_m;
                    if (Bm != 0ull) {
                        const bool isb = __builtin_amdgcn_inverse_ballot_w64(Bm);
                        lenh = isb ? nl + 1u : lenh;
                        hcand = isb ? nh - 1u : hcand;
                        Hm |= Bm;
                        M |= Bm;
                        Zm = (Zm & ~Bm) | (Bm & (Zm >> 1));
                    }
                }
                const unsigned long long DOMm = (ballot(r1 + 1u > lenh) & (Rm >> 1)) | (ballot(r2 + 2u > lenh) & (Rm >> 2));  // Rm >> k: lane + k starts a run
                const unsigned long long drop = Hm & (DOMm | ballot(lenh < LZ_MINHASH)) & ~Rm;  // lanes with a run of their own stay candidates
                Hm &= ~(DOMm | ballot(lenh < LZ_MINHASH));
                M &= ~drop;
                lenh = __builtin_amdgcn_inverse_ballot_w64(Hm) ? lenh : 0u;
                if (M == 0ull) {
                    p += 64u;
                    own = lds_load6(in, p + lane);
                    pbv = (uint32_t)in[p + lane - 1u];
                    continue;
                }
            }
            const unsigned long long URm = ballot(lenr >= lenh);  // ties go to the run (offset 1)
            uint32_t len = lenr > lenh ? lenr : lenh;
            const uint32_t off = __builtin_amdgcn_inverse_ballot_w64(URm) ? 1u : pos - hcand;
            // "still matching": a run that reaches the window end, or a hash match alive after 20 bytes —
            // finished cooperatively if it is the window's last match
            const unsigned long long RFm = URm & ballot(lenr == to_end);
            const unsigned long long MOREm = Hm & Zm & ~URm;
            unsigned long long LNG = RFm | MOREm;
            if (p + 100u > matchlimit) {  // only the last windows of a stream can reach the end-of-block limits
                asm volatile("" ::: "memory");
                const uint32_t maxlen = matchlimit - pos;  // lanes past matchlimit are not in M
                LNG &= ballot(len < maxlen);               // clipped at the end of the stream: not "long"
                len = len < maxlen ? len : maxlen;
            }
            // ---- greedy parse: the only serial part (scalar unit; 8 scalar instructions + one v_readlane per match).
            //      Probe builds (-DLZ_PROBE_SALU / -DLZ_PROBE_VALU: 16 extra independent s_mov / v_add per window) cost
            //      +14 % / +6 % kernel time: a wave's own serial instruction stream is the limit.  A variant with a
            //      per-lane precomputed "candidates after my match" mask (4 scalar + 2 v_readlane per match) was
            //      slower (36.5 vs 35.4 ms): the readlane -> compare -> branch chain is longer than this one.
            // A match that is still matching after its per-lane cap is finished cooperatively below: in the default mode
            // only the window's last selected match (the loop ends on it), in the high-effort mode every selected one — a
            // capped match is NOT followed by its own continuation (the next candidate is the most recent place with the
            // next 12 bytes, usually somewhere else), so each cut costs a sequence: +3.9 % ratio for +15 % time.
            // ("+&s": the loop rewrites m while it still needs M — without the early-clobber mark the two, equal on entry,
            //  may be given the same register pair)
            unsigned long long SEL = 0ull, mrest = M;
            uint32_t lcur, last;
            for (;;) {
                if constexpr (LONGRUN) {   // high-effort mode: leave the loop at every selected lane of LNG
                    asm volatile("1:\n\t"
                                 "s_ff1_i32_b64 %[last], %[m]\n\t"
                                 "s_bitset1_b64 %[sel], %[last]\n\t"
                                 "v_readlane_b32 %[lcur], %[len], %[last]\n\t"
                                 "s_add_i32 %[lcur], %[lcur], %[last]\n\t"
                                 "s_bitcmp1_b64 %[lng], %[last]\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_cmp_gt_u32 %[lcur], 63\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_lshl_b64 %[m], -1, %[lcur]\n\t"
                                 "s_and_b64 %[m], %[m], %[M]\n\t"
                                 "s_cbranch_scc1 1b\n"
                                 "2:"
                                 : [sel] "+&s"(SEL), [m] "+&s"(mrest), [last] "=&s"(last), [lcur] "=&s"(lcur)
                                 : [M] "s"(M), [len] "v"(len), [lng] "s"(LNG)
                                 : "scc");
                } else {                   // default: 8 scalar instructions + one v_readlane per match; only the last match is finished
                    asm volatile("1:\n\t"
                                 "s_ff1_i32_b64 %[last], %[m]\n\t"
                                 "s_bitset1_b64 %[sel], %[last]\n\t"
                                 "v_readlane_b32 %[lcur], %[len], %[last]\n\t"
                                 "s_add_i32 %[lcur], %[lcur], %[last]\n\t"
                                 "s_cmp_gt_u32 %[lcur], 63\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_lshl_b64 %[m], -1, %[lcur]\n\t"
                                 "s_and_b64 %[m], %[m], %[M]\n\t"
                                 "s_cbranch_scc1 1b\n"
                                 "2:"
                                 : [sel] "+&s"(SEL), [m] "+&s"(mrest), [last] "=&s"(last), [lcur] "=&s"(lcur)
                                 : [M] "s"(M), [len] "v"(len)
                                 : "scc");
                }
                if (!((LNG >> last) & 1ull)) break;
                // the selected match is still matching (20-byte cap, or a run that reaches the window end): finish it
                LZ_STAT(4, 1);
                const uint32_t ps = p + last;
                const uint32_t c = ps - (uint32_t)__builtin_amdgcn_readlane((int)off, (int)last);
                uint32_t ml = lcur - last;
                // most tails are short: first look at the next 64 bytes, one byte per lane
                const uint32_t kb = ml + lane;
                const uint32_t room = matchlimit - (ps + ml);  // > 0: a match that reached matchlimit is not in LNG
                const unsigned long long neb = ballot(in[ps + kb] != in[c + kb]) | (room < 64u ? ~0ull << room : 0ull);
                if (neb != 0ull) {
                    ml += (uint32_t)__ffsll((long long)neb) - 1u;
                } else {
                    LZ_STAT(5, 1);
                    ml += 64u;
                    for (;;) {  // a long run: 256 bytes per step
                        const uint32_t k = ml + 4u * lane;
                        uint32_t nm = 0u;
                        if (ps + k < matchlimit) {
                            nm = eq_bytes(lds_load4(in, ps + k) ^ lds_load4(in, c + k));
                            const uint32_t room = matchlimit - (ps + k);
                            nm = nm < room ? nm : room;
                        }
                        const unsigned long long stop = ballot(nm < 4u);
                        if (stop == 0ull) {
                            ml += 256u;
                            continue;
                        }
                        const uint32_t f2 = (uint32_t)__ffsll((long long)stop) - 1u;
                        ml += 4u * f2 + (uint32_t)__builtin_amdgcn_readlane((int)nm, (int)f2);
                        break;
                    }
                }
                len = lane == last ? ml : len;
                lcur = last + ml;
                if (LONGRUN && ((RFm >> last) & 1ull) && ml + 1u > best_len) {  // a longer run of one byte value: bytes ps-1 .. ps+ml-1
                    best_len = ml + 1u;
                    best_start = ps - 1u;
                    best_byte = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)last) & 0xFFu;
                }
                if (!LONGRUN || lcur >= 64u) break;   // (default mode: that was the window's last match)
                mrest = M & (~0ull << lcur);
                if (mrest == 0ull) break;
            }
            // next window's own bytes: request now, consumed after the enqueue below
            const uint32_t np = p + (lcur > 64u ? lcur : 64u);
#ifdef LZ_PROBE_SALU  // sensitivity probe (development builds only): 16 extra scalar instructions per window
            asm volatile(".rept 16\n\ts_mov_b32 s100, 0\n\t.endr" ::: "s100");
#endif
#ifdef LZ_PROBE_VALU  // ... or 16 extra vector instructions per window
            {
                uint32_t pv = lane;
                asm volatile(".rept 16\n\tv_add_u32 %0, 1, %0\n\t.endr" : "+v"(pv));
            }
#endif
            const Own6 nown = lds_load6(in, np + lane);
            const uint32_t npbv = (uint32_t)in[np + lane - 1u];
            // ---- enqueue the selected matches; layout + emission happen once per ~12 windows (flush), one
            //      sequence per lane, instead of once per window with 4 of 64 lanes busy
            {
                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(SEL >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)SEL, 0u));
                const uint32_t addr = (queue_lds + qn * 8u) + slot * 8u;  // scalar base + lane slot
                const unsigned long long ent = (unsigned long long)(pos | (len << 16)) | ((unsigned long long)off << 32);
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved)
                             : "s"(SEL), "v"(addr), "v"(ent)
                             : "memory");
            }
            qn += (uint32_t)__popcll(SEL);
            LZ_STAT(7, __popcll(SEL));
            if (qn > 48u) {
                LZ_STAT(6, 1);
                lz4_flush_queue(in, out, queue, qn, op, anchor, lane);
                qn = 0u;
            }
            p = np;
            own = nown;
            pbv = npbv;
        }
    }
    if (qn) lz4_flush_queue(in, out, queue, qn, op, anchor, lane);
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
// MW = minimum waves per SIMD the register allocator must leave room for (0: no constraint, workgroups of up
// to 16 waves for typesize 16).  The common genotype case (typesize 2 -> 2-wave workgroups) is latency bound,
// so it is compiled for 8 resident waves per SIMD (<= 64 VGPRs).
// PLANES: the blocks exist as bit planes (phase A generates the bytes); a template parameter so that the int8
// instantiations keep their registers (the default one sits at the 72-register line of 7 waves per SIMD)
template <int MW, int ALGO, bool PLANES>
__global__ __launch_bounds__(MW ? 128 : 1024, MW ? MW : 1) void k_lz4_blocks(const uint8_t *__restrict__ src, uint32_t nblocks,
                                                     uint64_t chunk_nbytes, uint32_t typesize, uint32_t blocksize,
                                                     uint32_t split, uint32_t sstride, uint32_t hashlog, uint32_t algo,
                                                     uint8_t *__restrict__ scratch, uint64_t slot_bytes,
                                                     uint32_t *__restrict__ csize, uint32_t n_total, const uint8_t *__restrict__ planes,
                                                     PlanesGeom pg, const uint32_t *__restrict__ mark_flag, uint32_t tag)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nwaves = blockDim.x >> 6;
    // scan mode: the bit-plane coders say whether they left anything (lz4bits.hip: the word holds this call's tag if so)
    if (mark_flag && __builtin_nontemporal_load(mark_flag) != tag) return;
    // algo bit 8: scan mode — only the streams the bit-plane encoders (lz4bits.hip) left marked (csize == 0xFFFFFFFF) are
    // coded: the (small, fixed) grid scans the stream sizes, 64 blocks per load, and works on the blocks that still hold one
    const bool only_marked = (algo & 0x100u) != 0u;
    for (uint32_t b0 = only_marked ? blockIdx.x * 64u : blockIdx.x; b0 < (only_marked ? n_total : blockIdx.x + 1u);
         b0 += only_marked ? gridDim.x * 64u : 1u) {
    unsigned long long todo = 1ull;
    if (only_marked) {   // every wave of the workgroup reads the same sizes and gets the same mask
        const uint32_t bb = b0 + (threadIdx.x & 63u);
        bool any = false;
        if (bb < n_total)
            for (uint32_t j = 0; j < nwaves; ++j) any = any || csize[(uint64_t)bb * nwaves + j] == 0xFFFFFFFFu;
        todo = __builtin_amdgcn_ballot_w64(any);
    }
    while (todo != 0ull) {   // (workgroup-uniform)
    const uint32_t bid = only_marked ? b0 + (uint32_t)__builtin_ctzll(todo) : b0;
    todo &= todo - 1ull;
    // wave index through readfirstlane: everything derived from it (stream base, output slot) stays in SGPRs,
    // so the byte stores below use the SGPR-base + 32-bit-offset addressing form
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint64_t chunk = bid / nblocks;
    const uint32_t b = bid - (uint32_t)(chunk * nblocks);
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
    if (PLANES) {
        // the block exists as bit planes (include/hhgt.h "Bit-plane form"; typesize 2, 8 KiB blocks): the shuffled byte
        // planes are generated from the bits; bytes of calls beyond 0 / 1 / missing come from their place in src
        uint64_t pcol;
        uint32_t prow, pbi;
        planes_block(pg, bid, &pcol, &prow, &pbi);
        const uint64_t kst = (uint64_t)pg.S_pad * 32ull;   // bytes between the kind-planes of a tile
        for (uint32_t i = threadIdx.x; i < 512u; i += blockDim.x) {   // byte i of each plane = variants 8 i .. 8 i + 7
            const uint8_t *pl = planes + planes_piece(pg, pcol, pbi * 16u + (i >> 5), 0u, prow) + (i & 31u);
            const uint32_t o[2] = {pl[0], pl[kst]}, e[2] = {pl[2ull * kst], pl[3ull * kst]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t w[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    uint32_t v = (o[h] >> k) & 1u;
                    if ((e[h] >> k) & 1u) v = v ? 0xF7u : (src ? (uint32_t)blk[(8u * i + (uint32_t)k) * 2u + (uint32_t)h] : 0u);
                    w[k >> 2] |= v << (8 * (k & 3));
                }
                *reinterpret_cast<uint2 *>(data + (uint32_t)h * pstride + 8u * i) = make_uint2(w[0], w[1]);
            }
        }
    } else if (typesize == 1u) {
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
    {   // LZ_SLACK: what the lanes see behind the end of a stream is zeros, not a neighbour's table or a previous block
        const uint32_t w_ = threadIdx.x >> 6;
        if (w_ < nstreams)
            for (uint32_t k = threadIdx.x & 63u; k < LZ_SLACK; k += 64u) data[(size_t)w_ * pstride + neblock + k] = 0;
    }
    __syncthreads();
    // ---- phase B: one wave per stream
    if (only_marked && csize[(uint64_t)bid * nwaves + wave] != 0xFFFFFFFFu) {
        // this stream of the block was coded by the bit-plane encoder
    } else if (wave < nstreams) {
        const uint8_t *in = data + (size_t)wave * pstride;
        const uint64_t sidx = (uint64_t)bid * nwaves + wave;
        uint8_t *out = scratch + sidx * slot_bytes;
        uint16_t *tb = tabs + (size_t)wave * ((1u << hashlog) + 2u);
        // sequence queue: 64 entries per wave, 8-aligned after the tables (offset arithmetic keeps the LDS address space)
        const uint32_t qoff = (nwaves * sstride + 16u + nwaves * ((2u << hashlog) + 4u) + 7u) & ~7u;
        uint2 *queue = reinterpret_cast<uint2 *>(smem + qoff) + wave * 64u;
        uint32_t cs = ALGO == 1   ? lz4_wave_compress(in, neblock, tb, hashlog, out)
                      : ALGO == 5 ? lz4_wave_compress_v6<true, false>(in, neblock, tb, hashlog, out, queue)
                      : ALGO == 7 ? lz4_wave_compress_v6<false, true>(in, neblock, tb, hashlog, out, queue)
                                  : lz4_wave_compress_v6<false, false>(in, neblock, tb, hashlog, out, queue);
        (void)algo;
        if (cs >= neblock) {  // incompressible: Blosc stores the (shuffled) stream verbatim
            for (uint32_t k = lane; k < neblock; k += 64u) out[k] = in[k];
            cs = neblock;
        }
        if (lane == 0) csize[sidx] = cs;
    } else if (lane == 0) {
        csize[(uint64_t)bid * nwaves + wave] = 0u;
    }
    if (only_marked) __syncthreads();   // the next marked block reuses the LDS
    }
    }
}

int launch_lz4_blocks(const uint8_t *d_src, const uint8_t *d_planes, PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize,
                      int blocksize, uint8_t *d_scratch, size_t slot_bytes, uint32_t *d_csize, int clevel, uint32_t *d_flags, uint32_t tag,
                      hipStream_t st)
{
    const uint32_t *mark_flag = nullptr;   // scan mode: where the last bit-plane coder of this call says whether it left marks
    if (d_planes && (typesize != 2 || blocksize != 8192 || chunk_nbytes % 8192 || (reinterpret_cast<uintptr_t>(d_planes) & 15u))) {
        hhgt_set_error("lz4: bit planes stand for typesize 2, 8 KiB blocks, chunks of whole blocks, 16-byte aligned");
        return HHGT_ERR_ARG;
    }
    const int fast = clevel <= 2 ? 1 : (clevel >= 7 ? 2 : 0);
    // the bit-plane encoder (lz4bits.hip) takes the case the path is built for — typesize 2, 8 KiB blocks, default
    // effort — and marks the streams it cannot code (a byte > 1, very dense planes); this kernel then only runs those.
    // HHGT_LZ4_BITPLANES=0 keeps everything on the byte-wise encoder.
    static const bool bp_env = !(getenv("HHGT_LZ4_BITPLANES") && atoi(getenv("HHGT_LZ4_BITPLANES")) == 0);
    const bool bitplanes = bp_env && typesize == 2 && blocksize == 8192 && chunk_nbytes % 8192 == 0 &&
                           (d_planes || (reinterpret_cast<uintptr_t>(d_src) & 15u) == 0) && slot_bytes >= 4128;
    if (bitplanes) {
        // effort: candidates tried per one along the hash chain (clevel 1-2: none, offset-1 runs only)
        // clevel 9 — what the file-writing paths run at — also parses lazily (+ 0x100; HHGT_LZ4_LAZY=0 / 1 overrides: 1 at the
        // default depth is the measurement of what the rule costs the bench)
        static const int depth_env = getenv("HHGT_LZ4_DEPTH") ? atoi(getenv("HHGT_LZ4_DEPTH")) : -1;
        static const int lazy_env = getenv("HHGT_LZ4_LAZY") ? atoi(getenv("HHGT_LZ4_LAZY")) : -1;
        int depth = depth_env >= 0 ? depth_env : (clevel <= 2 ? 0 : clevel <= 4 ? 1 : clevel <= 6 ? 2 : clevel == 7 ? 4 : clevel == 8 ? 8 : 12);
        if ((lazy_env < 0 ? clevel >= 9 && depth == 12 : lazy_env != 0) && (depth == 2 || depth == 12)) depth |= 0x100;
        bool exc_ran = false;
        const int rc = launch_lz4_bitplanes(d_planes ? d_planes : d_src, d_planes != nullptr, pg, n_chunks * (chunk_nbytes / 8192), d_scratch,
                                            slot_bytes, d_csize, depth, d_flags, tag, &exc_ran, st);
        if (rc != HHGT_OK) return rc;
        if (d_flags) mark_flag = d_flags + (exc_ran ? 1 : 0);
    }
    const uint32_t split = (typesize >= 2 && typesize <= 16 && blocksize / typesize >= 128) ? 1u : 0u;
    const uint32_t nwaves = split ? (uint32_t)typesize : 1u;
    const uint32_t nblocks = (uint32_t)((chunk_nbytes + blocksize - 1) / blocksize);
    // per-stream LDS stride: stream bytes + LZ_SLACK zero bytes for the lanes' lookahead, 16-byte aligned
    const uint32_t max_stream = split ? (uint32_t)blocksize / (uint32_t)typesize : (uint32_t)blocksize;
    uint32_t sstride = (max_stream + LZ_SLACK + 15u) & ~15u;
    if (split && chunk_nbytes % blocksize) {
        // a leftover block is compressed as ONE stream laid out contiguously over the data area
        uint32_t need = ((uint32_t)(chunk_nbytes % blocksize) + LZ_SLACK + 15u) & ~15u;
        if (need > sstride * nwaves) sstride = (need + nwaves - 1) / nwaves;
        sstride = (sstride + 15u) & ~15u;
    }
    const size_t data_bytes = (size_t)nwaves * sstride + 16u;
    // hash table size: the kernel is latency/issue bound, so resident waves matter more than table
    // reach.  Take the largest table (<= 4096 entries) that does not cost a resident workgroup
    // relative to the smallest one (512 entries); 160 KiB LDS and 32 waves per CU.
    static const int hl_env = getenv("HHGT_LZ4_HASHLOG") ? atoi(getenv("HHGT_LZ4_HASHLOG")) : 0;
    auto occ = [&](uint32_t hl) {
        size_t l = data_bytes + (size_t)nwaves * ((2u << hl) + 4u) + 8u + (size_t)nwaves * 512u;
        size_t by_lds = (160 * 1024) / l, by_waves = 32 / nwaves;
        return by_lds < by_waves ? by_lds : by_waves;
    };
    uint32_t hashlog = 12;
    // (HHGT_LZ4_MINWAVES=8 HHGT_LZ4_HASHLOG=7 gives 16 workgroups per CU: LZ4 alone 35.3 -> 34.5 ms, but the whole
    //  two-stream step does not move (53.3 ms) because the index kernel then finds no room beside it; on genotype
    //  planes the ratio is the same 4.326 with 64 .. 512 table entries)
    while (hashlog > 9 && occ(hashlog) < occ(9)) --hashlog;
    if (hl_env >= 6 && hl_env <= 13) hashlog = (uint32_t)hl_env;
    // HHGT_LZ4_LDSPAD: extra dynamic LDS per workgroup — an experiment knob that caps how many LZ4 workgroups
    // share a CU, leaving LDS for the HBM-bound kernels of the other stream
    static const size_t lds_pad = getenv("HHGT_LZ4_LDSPAD") ? (size_t)atoi(getenv("HHGT_LZ4_LDSPAD")) : 0;
    const size_t lds = data_bytes + (size_t)nwaves * ((2u << hashlog) + 4u) + 8u + (size_t)nwaves * 512u + lds_pad;
    if (lds > 160 * 1024 - 64) {
        hhgt_set_error("lz4: block of %d bytes x typesize %d does not fit LDS", blocksize, typesize);
        return HHGT_ERR_ARG;
    }
    static size_t attr_lds = 64 * 1024;  // dynamic LDS above 64 KiB needs an explicit opt-in
    if (lds > attr_lds) {
        const void *fns[] = {reinterpret_cast<const void *>(k_lz4_blocks<0, 1, false>), reinterpret_cast<const void *>(k_lz4_blocks<0, 5, false>),
                             reinterpret_cast<const void *>(k_lz4_blocks<8, 5, false>), reinterpret_cast<const void *>(k_lz4_blocks<0, 6, false>),
                             reinterpret_cast<const void *>(k_lz4_blocks<7, 6, false>), reinterpret_cast<const void *>(k_lz4_blocks<8, 6, false>),
                             reinterpret_cast<const void *>(k_lz4_blocks<0, 7, false>), reinterpret_cast<const void *>(k_lz4_blocks<7, 7, false>),
                             reinterpret_cast<const void *>(k_lz4_blocks<8, 5, true>), reinterpret_cast<const void *>(k_lz4_blocks<7, 6, true>),
                             reinterpret_cast<const void *>(k_lz4_blocks<7, 7, true>)};
        for (const void *f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    // HHGT_LZ4_ALGO: 6 = window-parallel encoder with batched emission (default), 1 = the simple first version (A/B)
    static const uint32_t algo_env = getenv("HHGT_LZ4_ALGO") ? (uint32_t)atoi(getenv("HHGT_LZ4_ALGO")) : 6u;
    const uint32_t algo = algo_env | (bitplanes ? 0x100u : 0u);
    uint64_t grid = n_chunks * nblocks;
    if (grid == 0) return HHGT_OK;
    // scan mode: a fixed grid (enough workgroups to fill the chip) scans the stream sizes, 64 blocks per workgroup and step
    if (bitplanes) grid = (grid + 63) / 64 < 256u * 14u ? (grid + 63) / 64 : 256u * 14u;
    if (grid > 0x7fffffffull) {
        hhgt_set_error("lz4: too many blocks");
        return HHGT_ERR_ARG;
    }
    static const int mw_env = getenv("HHGT_LZ4_MINWAVES") ? atoi(getenv("HHGT_LZ4_MINWAVES")) : 7;
    const int mw = nwaves <= 2 ? mw_env : 0;
    static const bool dbg = getenv("HHGT_LZ4_DEBUG") != nullptr;
    if (dbg) {  // development: what the runtime thinks fits on a CU
        int n7 = -1, n8 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n7, k_lz4_blocks<7, 6, false>, (int)(64u * nwaves), lds);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n8, k_lz4_blocks<8, 6, false>, (int)(64u * nwaves), lds);
        fprintf(stderr, "[hhgt lz4] nwaves=%u hashlog=%u lds=%zu B/workgroup, workgroups per CU: <7,6> %d  <8,6> %d\n", nwaves, hashlog,
                lds, n7, n8);
    }
#define LZ_LAUNCH2(MWV, ALG, PL)                                                                                      \
    hipLaunchKernelGGL((k_lz4_blocks<MWV, ALG, PL>), dim3((uint32_t)grid), dim3(64u * nwaves), lds, st, d_src, nblocks, \
                       chunk_nbytes, (uint32_t)typesize, (uint32_t)blocksize, split, sstride, hashlog, algo, d_scratch, \
                       (uint64_t)slot_bytes, d_csize, (uint32_t)(n_chunks * nblocks), d_planes, pg, mark_flag, tag)
#define LZ_LAUNCH(MWV, ALG) LZ_LAUNCH2(MWV, ALG, false)
    // effort: 1 = run candidate only (clevel 1-2), 0 = hash + run candidates (clevel 3-6, the default 5), 2 = plus the
    // long-run source candidate (clevel 7-9)
    if (d_planes) {   // typesize 2: two-wave workgroups
        if (fast == 1) LZ_LAUNCH2(8, 5, true);
        else if (fast == 2) LZ_LAUNCH2(7, 7, true);
        else LZ_LAUNCH2(7, 6, true);
    } else if ((algo & 0xFFu) == 1u) LZ_LAUNCH(0, 1);
    else if (fast == 1 && nwaves <= 2) LZ_LAUNCH(8, 5);
    else if (fast == 1) LZ_LAUNCH(0, 5);
    else if (fast == 2 && nwaves <= 2) LZ_LAUNCH(7, 7);
    else if (fast == 2) LZ_LAUNCH(0, 7);
    else if (mw == 8) LZ_LAUNCH(8, 6);
    else if (mw == 7) LZ_LAUNCH(7, 6);
    else LZ_LAUNCH(0, 6);
#undef LZ_LAUNCH2
#undef LZ_LAUNCH
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
