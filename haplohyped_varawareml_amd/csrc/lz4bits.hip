// lz4bits.hip — LZ4 block encode of 0/1 byte planes by walking the ONES, not the bytes (gfx950).
//
// Same place in the path as lz4.hip (the shuffle + LZ4 step of the HDF5 filter the reference invokes at
// /root/reference/src/haplohyped/vcf_to_h5.py:134-135), same output contract (a valid LZ4 block stream per Blosc
// stream, "decompresses to identical bytes"), for the case the path is built for: typesize 2, 8 KiB Blosc blocks,
// i.e. two 4 KiB byte planes per block, each the haplotype of one sample over 4096 variants — bytes that are 0 or 1
// and about 94 % zeros.  The byte-wise encoder in lz4.hip spends ~3.8 k vector + ~3.3 k scalar instructions per plane
// looking at every one of the 4096 positions (64 per window, one per lane); it is bound by instruction issue with HBM
// at 6 %.  This kernel turns the plane into a 4096-bit map and works on the LIST OF ONES (~260 per plane):
//
//   * one lane per one, 64 ones per window (4-5 windows per plane instead of 52-64);
//   * candidate table keyed on the GAP behind the one (distance to the next one, clipped to 40: 64 entries, no hash),
//     updated by ONE ds_wrxchg_rtn_b32 per window: the LDS serves lanes that hit the same address in ascending lane
//     order (tools/micro/lds_xchg_order.hip, checked by a test), so every lane gets the most recent earlier one with its
//     key — exact sequential table semantics, 64 insertions at once.  (The first version keyed the table on the 12 bits
//     at the one, LZ4's habit; on these planes that context is "a one and eleven zeros" nearly everywhere.  An equal
//     first gap is what a match of ones needs to get past its first step: ratio 5.31 -> 5.86 at two candidates.)
//   * match lengths come from comparing GAPS between ones, not bytes: equal gaps, then one plus the shorter of the
//     first unequal pair.  The gaps also exist as BYTES (clipped to 255), so a candidate is judged by one unaligned
//     4-byte fetch, an xor and a count-trailing-zeros (three gaps + the open one); the longest agreement along the chain
//     wins — the zeros the two have in common IN FRONT count as well: a match is pulled back over them (up to 64) —
//     (picking by length packs tighter than picking by the local saving: the parse is greedy) and only the winner
//     is worked out exactly from the positions, extended past the third gap where it still agrees, pulled back over
//     the zeros in front, and priced;
//   * every one decides locally between "match" and "literal" by cost (3 bytes per sequence against the literals and
//     runs it replaces) and thereby where the next coded one is: the greedy parse is a linked list nxt(j), followed
//     inside a window by pointer doubling (6 rounds), not by a serial loop;
//   * behind whatever a coded one ends with, the zeros up to the next one go out as an offset-1 run — if there are
//     four that the NEXT coded one's match does not pull back over anyway (decided at layout time, where the next
//     coded one is the neighbouring queue entry; round 3: run rule and an 8-zero pull-back that knew nothing of
//     each other packed 8 % looser with the same candidates);
//   * every coded one lays out and writes its (at most two) sequences itself, literals generated from the bit map.
//
// tools/sim/gapenc_ref.c states the same algorithm on the CPU, decision for decision; the kernel's streams are
// compared with it byte for byte (tests/test_gpu_lz4_bitplanes.py) and decoded by liblz4 / the oracle like every
// other stream.  Planes with a byte > 1 (missing calls, -9) or more than 636 ones are left to the byte-wise
// encoder (csize = MARK; lz4.hip's kernel then runs in "marked streams only" mode).
#include "common.h"

#define BP_N 4096
#ifndef BP_SELECT_SCALAR
#define BP_SELECT_SCALAR 0   // which ones of a window the parse visits: 0 pointer doubling through LDS; 1 walked on the scalar unit
                             // (built again in round 4, leaner than round 2's: one v_readlane + ~6 scalar instructions per coded
                             // one, no LDS — byte-identical streams, LZ4 stage 14.0 against 12.1 ms: a taken branch and a
                             // vector -> scalar hand-over per hop are a longer chain than twelve LDS round trips per window)
#endif
#ifndef BP_MAXONES
#define BP_MAXONES 636
#endif
#ifndef BP_STAGE
#define BP_STAGE 576    // staging area; with the queue below the workgroup stays under 11 KiB of LDS = 14 workgroups per CU
#endif
#ifndef BP_WAVES
#define BP_WAVES 7       // waves per SIMD the plain instantiations are compiled for (LDS must allow 2 x BP_WAVES workgroups per CU)
#endif
#define BP_QCAP 100     // queued coded ones a wave can hold (a window adds at most 64 to fewer than 64)
// the exception-aware instantiation carries a second bit map and the ones' classes (624 bytes per wave): its list and its staging
// area are smaller by as much, so that it too stays under 11 KiB per workgroup = 14 workgroups per CU, 7 waves per SIMD (round 4;
// it ran at 13 / 6 before, and this kernel's time is 1 / occupancy)
#ifndef BP_MAXONES_EXC
#define BP_MAXONES_EXC 540
#endif
#ifndef BP_STAGE_EXC
#define BP_STAGE_EXC 448
#endif
#ifndef BP_WAVES_EXC
#define BP_WAVES_EXC 7
#endif
template <bool EXC> struct BpCfg {
    static constexpr uint32_t MAXONES = EXC ? BP_MAXONES_EXC : BP_MAXONES;
    static constexpr uint32_t STAGE = EXC ? BP_STAGE_EXC : BP_STAGE;
    static constexpr uint32_t PTRASH = MAXONES + 7;   // P's spare entry: written by lanes that have nothing to store, never read
};
#define BP_HLOG 6
#define BP_GAPCLIP 40
#define BP_MINM 6     // total length a hash match must have
#define BP_TMIN 4     // zeros an offset-1 run must cover to be worth its 3 bytes
#define BP_BACK 64    // zeros in front of the one a match is pulled back over (round 2: 8)
#define BP_STEPS 16   // exact extension of the chosen candidate
#define BP_PICK 3     // full gaps the pick looks at (one dword of gap bytes)
#define BP_LAZY_MIN 3 // LAZY: bytes the next one's match must reach further, on top of what giving up costs
#ifdef BP_MARKS   // development: section markers in the ISA listing (hipcc -S -DBP_MARKS), tools/isa_sections.py counts per section
#define BP_MARK(name) asm volatile("; ==MARK " name)
#else
#define BP_MARK(name)
#endif
#ifndef BP_SKIP
#define BP_SKIP 0   // development: 1 no emission, 2 no layout either, 3 no selection, 4 no window loop, 5 phase A only (invalid output; timing only)
#endif
#define BP_MFLIMIT (BP_N - 12)
#define BP_MATCHLIMIT (BP_N - 5)

template <bool CHAIN, bool EXC> struct BpLds {
    uint32_t bm[132];                // bit map of the plane's nonzero bytes: 128 dwords + zero padding
    uint32_t xm[EXC ? 132 : 1];      // EXC: which of them are 0xF7 (missing calls)
    uint32_t cls[EXC ? 24 : 1];      // EXC: the same per ONE: bit i = the one with P-index i is a missing call
    uint32_t tab[1 << BP_HLOG];      // min(gap behind the one + 1, BP_GAPCLIP) -> one index + 1
    uint8_t flag[72];                // pointer-doubling marks of a window ([0, 64]; [68]: where lanes with nothing to mark write)
    uint16_t wpre[64];               // ones in front of bit-map word w (64-bit words)
    uint16_t P[BpCfg<EXC>::MAXONES + 8];      // P[j + 1] = q_j + 1 (P[0] = 0: a virtual one at -1; P[m + 1] = P[m + 2] = n + 1)
    uint16_t chain[CHAIN ? BpCfg<EXC>::MAXONES + 8 : 4];   // chain[j + 1] = (previous one with the same context hash) + 1, 0 = none
    uint32_t gbw[(BpCfg<EXC>::MAXONES + 16) / 4];   // gap bytes: zeros behind the one with P-index i, clipped to 255 (255 from the last one on)
    uint2 queue[BP_QCAP + 1];        // coded ones waiting for layout + emission (bp_emit_batch takes 64 at a time); [BP_QCAP]: written by lanes that queue nothing
    uint8_t stage[BpCfg<EXC>::STAGE + 8];     // output staged here, written out in coalesced dwords; [BP_STAGE]: written by lanes that have no byte to store
};

// a queued coded one: what its sequences need that does not depend on the sequences in front of it
//   x: E (13 bits) | msr = q - nb (13) << 13 | onM << 26 | onT << 27 | (rs - E) << 28        (E: end of what the one codes;
//      onT: the run is long enough — whether it is emitted is decided against the next entry, bp_emit_batch)
//   y: re (13 bits) | off << 13                                                                (re: start of the next coded one)
__device__ __forceinline__ uint2 bp_entry(uint32_t E, uint32_t msr, bool onM, bool onT, uint32_t rs, uint32_t re, uint32_t off)
{
    return make_uint2(E | ((msr & 0x1FFFu) << 13) | (onM ? 1u << 26 : 0u) | (onT ? 1u << 27 : 0u) | ((rs - E) << 28), re | (off << 13));
}

// wave-wide inclusive scans on DPP row shifts (no LDS round trips): sum and max of non-negative values
__device__ __forceinline__ uint32_t bp_row_shr(uint32_t x, int d)
{
    switch (d) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    }
}

__device__ __forceinline__ uint32_t bp_scan_sum(uint32_t x, uint32_t lane)   // inclusive
{
    x += bp_row_shr(x, 1);
    x += bp_row_shr(x, 2);
    x += bp_row_shr(x, 4);
    x += bp_row_shr(x, 8);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31) + t0;
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + t1;
    const uint32_t row = lane >> 4;
    return x + (row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t1 : t2)));
}

__device__ __forceinline__ uint32_t bp_max(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t bp_scan_max(uint32_t x, uint32_t lane)   // inclusive
{
    x = bp_max(x, bp_row_shr(x, 1));
    x = bp_max(x, bp_row_shr(x, 2));
    x = bp_max(x, bp_row_shr(x, 4));
    x = bp_max(x, bp_row_shr(x, 8));
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
    const uint32_t t1 = bp_max((uint32_t)__builtin_amdgcn_readlane((int)x, 31), t0);
    const uint32_t t2 = bp_max((uint32_t)__builtin_amdgcn_readlane((int)x, 47), t1);
    const uint32_t row = lane >> 4;
    return bp_max(x, row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t1 : t2)));
}

// 32 bits of the map starting at bit q (0 <= q < 4096 + 96)
__device__ __forceinline__ uint32_t bp_bits(const uint32_t *bm, uint32_t q)
{
    const uint32_t i = q >> 5;
    const uint32_t lo = bm[i], hi = bm[i + 1];
    return __builtin_amdgcn_alignbit(hi, lo, q & 31u);
}

// four gap bytes starting at byte i of the gap array (two aligned dwords, one funnel shift)
__device__ __forceinline__ uint32_t bp_gap4(const uint32_t *gbw, uint32_t i)
{
    const uint32_t w = i >> 2;
    return __builtin_amdgcn_alignbit(gbw[w + 1u], gbw[w], (i & 3u) << 3);
}

// extension bytes of an LZ4 length field: x >= 15 ? (x - 15) / 255 + 1 : 0 = (x + 240) / 255 for every x; the division as
// a 24-bit multiply and a shift (exact below 65536; the compiler's v_mul_hi_u32 runs at a quarter of the rate, and the
// kernel evaluates this eight times per window)
__device__ __forceinline__ uint32_t bp_len_ext(uint32_t x) { return __umul24(x + 240u, 32897u) >> 23; }

// literal bytes of positions whose map bits are b (and, EXC, whose missing-call bits are x): 0, 1 or 0xF7
template <bool EXC> __device__ __forceinline__ uint32_t bp_lit(uint32_t b, uint32_t x, uint32_t k)
{
    const uint32_t v = (b >> k) & 1u;
    return EXC ? (((x >> k) & 1u) ? 0xF7u : v) : v;
}

#ifndef BP_EMITSKIP
#define BP_EMITSKIP 0   // development (timing only, invalid streams): 1 no literal bytes, 2 no long literal runs, 3 no sequence bytes at all
#endif
// one LZ4 sequence: literals [anchor, start) (bytes generated from the bit map), match (len, off).  OUT is a pointer
// into the staging area (LDS: ds_write_b8) or, for a window too large for it, into the stream's slot in global memory.
// TR: out[tr] is a byte nobody reads (staging area only) — lanes with nothing to store write there instead of sitting out
// an exec-mask region: `if (p) store` costs two scalar instructions and often a branch, `store at (p ? a : tr)` one vector
// select, and scalar issue is what this kernel runs out of first (probe builds).
template <bool EXC, bool TR, typename OUT>
__device__ __forceinline__ void bp_put_seq(const uint32_t *bm, const uint32_t *xm, OUT out, uint32_t tr, uint32_t at, uint32_t anchor,
                                           uint32_t start, uint32_t len, uint32_t off, bool on, uint32_t lane)
{
    const uint32_t ll = on ? start - anchor : 0u, ml = len - 4u;
    const uint32_t llx = on ? bp_len_ext(ll) : 0u, mlx = on ? bp_len_ext(ml) : 0u;
    if (BP_EMITSKIP == 3) return;
    // Straight-line byte stores for what nearly every sequence has — token, at most one extension byte per length,
    // offset — and ONE wave-uniform branch for lengths that need more (>= 270).
    const uint32_t lit = at + 1u + llx, o = lit + ll;
    const uint32_t tok = ((ll < 15u ? ll : 15u) << 4) | (ml < 15u ? ml : 15u);
    if (TR) {
        out[on ? at : tr] = (uint8_t)tok;
        out[on ? o : tr] = (uint8_t)(off & 0xFFu);
        out[on ? o + 1u : tr] = (uint8_t)(off >> 8);
        out[llx != 0u ? at + 1u : tr] = (uint8_t)(llx == 1u ? ll - 15u : 255u);
        out[mlx != 0u ? o + 2u : tr] = (uint8_t)(mlx == 1u ? ml - 15u : 255u);
    } else {
        if (on) {
            out[at] = (uint8_t)tok;
            out[o] = (uint8_t)(off & 0xFFu);
            out[o + 1u] = (uint8_t)(off >> 8);
        }
        if (llx != 0u) out[at + 1u] = (uint8_t)(llx == 1u ? ll - 15u : 255u);
        if (mlx != 0u) out[o + 2u] = (uint8_t)(mlx == 1u ? ml - 15u : 255u);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(llx > 1u || mlx > 1u) != 0ull, 0)) {   // (wave-uniform, rare)
        uint32_t r = ll - 15u - 255u;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t k = 1; k < llx; ++k) {
            out[at + 1u + k] = (uint8_t)(k + 1u == llx ? r : 255u);
            r -= 255u;
        }
        r = ml - 15u - 255u;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t k = 1; k < mlx; ++k) {
            out[o + 2u + k] = (uint8_t)(k + 1u == mlx ? r : 255u);
            r -= 255u;
        }
    }
    // literals.  Most runs are 1-3 bytes ("1", "1 0"): up to 6 go out as byte stores of the owning lane.
    // Longer runs (the literals of all the ones in between that code nothing pile up in front of the next sequence)
    // are written by the whole wave, one run at a time — a lockstep per-lane loop would run max(ll) times for all.
    if (BP_EMITSKIP != 1 && __builtin_amdgcn_ballot_w64(ll != 0u) != 0ull) {
        const uint32_t b = bp_bits(bm, anchor), x = EXC ? bp_bits(xm, anchor) : 0u;
        const uint32_t nsm = ll <= 6u ? ll : 0u;   // bytes this lane stores itself
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k) {
            if (TR) out[k < nsm ? lit + k : tr] = (uint8_t)bp_lit<EXC>(b, x, k);
            else if (k < nsm) out[lit + k] = (uint8_t)bp_lit<EXC>(b, x, k);
        }
        unsigned long long big = BP_EMITSKIP == 2 ? 0ull : __builtin_amdgcn_ballot_w64(ll > 6u);
        while (big != 0ull) {
            const int l = __builtin_ctzll(big);
            big &= big - 1ull;
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)ll, l);
            const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)anchor, l);
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)lit, l);
            for (uint32_t k = lane; k < n; k += 64u) out[dst + k] = (uint8_t)bp_lit<EXC>(bp_bits(bm, src + k), EXC ? bp_bits(xm, src + k) : 0u, 0u);
        }
    }
}

// staged bytes [0, n) -> global, whole dwords coalesced, the last 1-3 bytes singly
__device__ __forceinline__ void bp_flush(const uint8_t *stage, uint8_t *__restrict__ dst, uint32_t n, uint32_t lane)
{
    struct __attribute__((packed)) PU32 {
        uint32_t v;
    };
    const uint32_t nd = n >> 2;
    for (uint32_t i = lane; i < nd; i += 64u) reinterpret_cast<PU32 *>(dst + 4u * i)->v = reinterpret_cast<const uint32_t *>(stage)[i];
    const uint32_t t = (nd << 2) + lane;
    if (t < n) dst[t] = stage[t];
}

#define BP_FENCE() asm volatile("" ::: "memory")
// layout + emission of the first n (<= 64) of the qn queued coded ones, one per lane, in stream order (n < qn unless the
// stream ends with entry n - 1: every entry sees its successor).  The end of a coded one's last sequence is where the
// next one's literals start (pe): the neighbouring lane's value, no scan.
struct BpOut {
    uint32_t prev_end, gop, sop;   // end of the last sequence; bytes written to global / staged
};   // (by value: a reference across the "memory" fences would pin the counters to scratch memory)

template <bool EXC, typename LDS>
__device__ __forceinline__ BpOut bp_emit_batch(LDS &S, const uint32_t *bm, uint8_t *__restrict__ out, uint32_t n, uint32_t qn, BpOut st, uint32_t lane)
{
    const uint32_t *xm = S.xm;
    uint32_t prev_end = st.prev_end, gop = st.gop, sop = st.sop;
    const bool have = lane < n;
    uint2 en = S.queue[lane];   // (lane < 64 <= BP_QCAP: inside the queue whatever n is)
    en.x = have ? en.x : 0u;
    en.y = have ? en.y : 0u;
    const uint32_t E = en.x & 0x1FFFu, msr = (en.x >> 13) & 0x1FFFu, rs = E + (en.x >> 28), re = en.y & 0x1FFFu, off = en.y >> 13;
    const bool onM = (en.x >> 26) & 1u;
    bool onT = (en.x >> 27) & 1u;
    {   // the run is dropped when the next coded one's match starts so far in front of its one that fewer than BP_TMIN
        // zeros are left to the run (the next coded one with a match IS the next entry: msr <= re says so)
        uint32_t nx = S.queue[lane + 1u].x;
        nx = lane + 1u < qn ? nx : 0u;
        const uint32_t nmsr = (nx >> 13) & 0x1FFFu;
        const bool nM = (nx >> 26) & 1u;
        uint32_t nbk = nM && nmsr <= re ? re - nmsr : 0u;
        const uint32_t zt = re - rs;
        nbk = nbk < zt ? nbk : zt;
        onT = onT && zt - nbk >= BP_TMIN;
    }
    // end of this one's last sequence.  An entry whose run was dropped and that has no match emits nothing and hands on
    // its predecessor's end — which is an emitting entry's: a run is only dropped in front of an entry WITH a match
    const uint32_t F0 = onT ? re : E;
    uint32_t Fp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)F0, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    Fp = lane ? Fp : prev_end;
    const uint32_t F = (onM || onT) ? F0 : Fp;
    uint32_t pe = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)F, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    pe = lane ? pe : prev_end;
    uint32_t ms = msr > pe ? msr : pe;                                 // the previous sequence may have taken some of the zeros in front
    ms = onM ? ms : 0u;
    const uint32_t llM = onM ? ms - pe : 0u, lenM = E - ms;
    const uint32_t pe2 = onM ? E : pe;
    const uint32_t llT = onT ? rs - pe2 : 0u, lenT = re - rs;
    const uint32_t szM = onM ? 3u + bp_len_ext(llM) + llM + bp_len_ext(lenM - 4u) : 0u;
    const uint32_t szT = onT ? 3u + bp_len_ext(llT) + llT + bp_len_ext(lenT - 4u) : 0u;
    const uint32_t sincl = bp_scan_sum(szM + szT, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
    BP_MARK("sizes_done");
    // the batch's bytes go to the staging area; what is staged leaves in coalesced dwords when the next batch would not
    // fit (a batch larger than the whole area is written to global memory directly)
    if (__builtin_expect(sop + total > BpCfg<EXC>::STAGE, 0)) {
        bp_flush(S.stage, out + gop, sop, lane);
        gop += sop;
        sop = 0;
    }
    const uint32_t at = sincl - (szM + szT);
    if (__builtin_expect(total > BpCfg<EXC>::STAGE, 0)) {   // (wave-uniform, rare: a batch with hundreds of literals)
        bp_put_seq<EXC, false>(bm, xm, out + gop, 0u, at, pe, ms, lenM, off, onM, lane);
        bp_put_seq<EXC, false>(bm, xm, out + gop, 0u, at + szM, pe2, rs, lenT, 1u, onT, lane);
        gop += total;
    } else {
        bp_put_seq<EXC, true>(bm, xm, S.stage + sop, BpCfg<EXC>::STAGE - sop, at, pe, ms, lenM, off, onM, lane);
        bp_put_seq<EXC, true>(bm, xm, S.stage + sop, BpCfg<EXC>::STAGE - sop, at + szM, pe2, rs, lenT, 1u, onT, lane);
        sop += total;
    }
    prev_end = (uint32_t)__builtin_amdgcn_readlane((int)F, (int)(n - 1u));
    return BpOut{prev_end, gop, sop};
}

// One plane, from its bit map(s) in registers to its stream (or its mark): everything behind the loads.  FLAG: which word of
// `flags` says "this call left a mark" for the launch that scans next; WRITE_MARK: the mark is written here (the scanning
// exception-aware launch finds it in place already).
template <int DEPTH, bool EXC, bool LAZY, int FLAG, bool WRITE_MARK, typename LDS>
__device__ __forceinline__ void bp_code_plane(LDS &S, uint32_t wlo, uint32_t whi, uint32_t xlo, uint32_t xhi, bool nonbinary, uint32_t bid,
                                              uint32_t wave, uint32_t lane, uint8_t *__restrict__ scratch, uint64_t slot_bytes,
                                              uint32_t *__restrict__ csize, uint32_t *__restrict__ flags, uint32_t tag)
{
    constexpr bool CHAIN = DEPTH > 1;
    uint32_t sink = 0;
    // ---- phase B: wave w codes plane w
    // The lanes of the wave exchange data through S with no barrier in between: the LDS executes a wave's instructions in
    // order.  What the COMPILER must not do is move a lane's read in front of another lane's earlier write; every such
    // hand-over below is an access whose address it cannot tell apart from the lane's own writes, and BP_FENCE marks
    // the phase boundaries for good measure.  (volatile pointers would also do — and turn every access into a
    // serialised flat load with its own wait, which made this kernel 2x slower than it had to be.)
    uint32_t *bm = S.bm;
    uint16_t *P = S.P;
    uint16_t *wpre = S.wpre;
    uint8_t *flag = S.flag;
    const uint64_t sidx = (uint64_t)bid * 2u + wave;
    uint8_t *out = scratch + sidx * slot_bytes;

    const uint32_t cnt = (uint32_t)__popc(wlo) + (uint32_t)__popc(whi);
    const uint32_t incl = bp_scan_sum(cnt, lane);
    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (nonbinary || m > BpCfg<EXC>::MAXONES) {
        // (the flag: read first — the line sits in the L2 and after the first few marking waves of an XCD it holds the tag)
        if (lane == 0 && flags && flags[FLAG] != tag) flags[FLAG] = tag;
        if (lane == 0 && WRITE_MARK) csize[sidx] = 0xFFFFFFFFu;   // left to the next coder (the scanning exception-aware launch finds the mark there already)
        return;
    }
    BP_MARK("scan_done");
    wpre[lane] = (uint16_t)(incl - cnt);
    BP_FENCE();
    {   // the list of ones
        // (the two halves of the lane's 64 positions one after the other: each round is a count-trailing-zeros, a clear
        // and a store, and there are as many rounds as the fullest 32-position word has ones — twice ~6 — where one loop
        // over the 64-bit word ran ~11 rounds of twice the work)
        // (no `if (lo != 0)` around the body: a lane that has run out stores to P's spare entry — an exec-mask region costs
        // two scalar instructions and a branch per round, and scalar issue is this kernel's tightest port)
        uint32_t lo = wlo, hi = whi, at = incl - cnt + 1u;
        P[lane == 0u ? 0u : BpCfg<EXC>::PTRASH] = 0;
        uint32_t base = 64u * lane + 1u;
        while (__builtin_amdgcn_ballot_w64(lo != 0u) != 0ull) {
            const bool on = lo != 0u;
            const uint32_t bpos = (uint32_t)__builtin_ctz(lo | 0x80000000u);
            if (EXC && on && ((xlo >> bpos) & 1u)) atomicOr(&S.cls[at >> 5], 1u << (at & 31u));   // (LDS: ds_or_b32)
            P[on ? at : BpCfg<EXC>::PTRASH] = (uint16_t)(base + bpos);
            at += on ? 1u : 0u;
            lo &= lo - 1u;
        }
        base += 32u;
        while (__builtin_amdgcn_ballot_w64(hi != 0u) != 0ull) {
            const bool on = hi != 0u;
            const uint32_t bpos = (uint32_t)__builtin_ctz(hi | 0x80000000u);
            if (EXC && on && ((xhi >> bpos) & 1u)) atomicOr(&S.cls[at >> 5], 1u << (at & 31u));
            P[on ? at : BpCfg<EXC>::PTRASH] = (uint16_t)(base + bpos);
            at += on ? 1u : 0u;
            hi &= hi - 1u;
        }
        P[lane < 4u ? m + 1u + lane : BpCfg<EXC>::PTRASH] = (uint16_t)(BP_N + 1);
    }
    BP_FENCE();
    if (DEPTH > 0) {   // gap bytes of all ones (a candidate is compared on them, 4 at a time)
        uint8_t *gb = reinterpret_cast<uint8_t *>(S.gbw);
        for (uint32_t i = lane; i < m + 8u; i += 64u) {
            const uint32_t i1 = i + 1u < BpCfg<EXC>::PTRASH ? i + 1u : BpCfg<EXC>::PTRASH;   // (i < m + 8 <= 644: the loads stay inside P)
            uint32_t g = (uint32_t)P[i1] - (uint32_t)P[i1 - 1u] - 1u;
            g = g < 255u ? g : 255u;
            gb[i] = (uint8_t)(i < m ? g : 255u);
        }
        BP_FENCE();
    }
#pragma unroll
    for (int k = 0; k < (1 << BP_HLOG) / 64; ++k) S.tab[64 * k + lane] = 0u;

    BP_MARK("plist_done");
    if (BP_SKIP >= 4) {
        if (lane == 0) csize[sidx] = P[m] & 1u;
        return;
    }
    uint32_t gop = 0, sop = 0, prev_end = 0, qn = 0;   // bytes written to global / staged; end of the last sequence; queued coded ones
    int cur = -1;   // next one to be coded (the virtual one in front of the stream first)
    for (int jw = -1; jw < (int)m; jw += 64) {
        BP_MARK("win_begin");
#ifdef BP_PROBE_SALU   // development: extra independent scalar / vector instructions per window — which issue port is the tight one?
#pragma unroll
        for (int pr = 0; pr < BP_PROBE_SALU; ++pr) asm volatile("s_mov_b32 s90, 0" ::: "s90");
#endif
#ifdef BP_PROBE_VALU
        {
            uint32_t pv;
#pragma unroll
            for (int pr = 0; pr < BP_PROBE_VALU; ++pr) asm volatile("v_mov_b32 %0, 0" : "=v"(pv));
        }
#endif
        const int j = jw + (int)lane;
        const bool valid = j < (int)m;
        const uint32_t jj = valid ? (uint32_t)(j + 1) : 0u;          // index into P of this one
        const uint32_t q1 = P[jj], qn1 = P[jj + 1u];
        const uint32_t qp1 = jj ? P[jj - 1u] : 0u;
        const int q = (int)q1 - 1;
        // ---- candidate table: keyed on the gap behind the one; exact recency through the LDS's lane order
        const bool can = DEPTH > 0 && valid && j >= 0 && q + 12 <= BP_N;
        uint32_t jc1 = 0;
        // EXC: classes of this one (bit 0) and of the four ones behind it
        const uint32_t c5a = EXC ? (bp_bits(S.cls, jj) & 31u) : 0u;
        if (can) {
            const uint32_t g1 = qn1 - q1;
            const uint32_t idx = (g1 < BP_GAPCLIP ? g1 : BP_GAPCLIP) ^ ((EXC && (c5a & 1u)) ? 63u : 0u);
            jc1 = atomicExch(&S.tab[idx], (uint32_t)(j + 1));
            if (CHAIN) S.chain[jj] = (uint16_t)jc1;   // what this one replaced: the next candidate down the chain
        }
        BP_FENCE();
        BP_MARK("hash_done");
        // ---- pick: the candidate with the longest forward agreement, judged on the gap BYTES — up to BP_PICK equal
        //      gaps, then 1 + the smaller of the next pair.  All a candidate costs is one unaligned 4-byte fetch from
        //      the gap bytes, an xor and a count-trailing-zeros; ties go to the nearer one.
        const uint32_t gq = q1 - qp1 - 1u;
        uint32_t len = 0, nb = 0, c1 = 0;
        bool hv = false;
        uint32_t a_last = jj;   // hv: P-index of the last one the match covers
        uint32_t z_last = 0;    // hv: zeros the match takes behind that one (0: it ends with the one)
        bool clamped = false;   // hv: the match was cut at the stream's match limit
        if (DEPTH > 0) {
            const uint32_t a4 = bp_gap4(S.gbw, jj);
            const uint8_t *gbb = reinterpret_cast<const uint8_t *>(S.gbw);
            const uint32_t fa = jj ? (uint32_t)gbb[jj - 1u] : 0u;    // zeros in front of this one (clipped to 255)
            const uint32_t na4 = ~a4;
            const uint32_t stop4 = ((na4 - 0x01010101u) & ~na4 & 0x80808080u) | 0xFF000000u;   // a 255 agrees with nothing; 3 gaps at most
            int best = -1;
            uint32_t bjq = 0, bk = 0;
#pragma unroll 1
            for (int dpt = 0; dpt < DEPTH; ++dpt) {
                const bool have = can && jc1 != 0u;
                if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
                const uint32_t jq = have ? jc1 : 1u;
                const uint32_t b4 = bp_gap4(S.gbw, jq);
                const uint32_t fb = (uint32_t)gbb[jq - 1u];
                uint32_t nextc = 0;
                if (CHAIN) nextc = S.chain[jq];
                const uint32_t x = (a4 ^ b4) | stop4;
                uint32_t k = (uint32_t)__builtin_ctz(x) >> 3;                           // agreeing gaps: 0..3
                bool same = true;
                if (EXC) {   // ... of which only those count whose next ones are of the same class; the first byte must agree at all
                    const uint32_t xc = c5a ^ (bp_bits(S.cls, jq) & 31u);
                    same = (xc & 1u) == 0u;
                    const uint32_t tv = (uint32_t)__builtin_ctz((xc >> 1) | 8u);
                    k = k < tv ? k : tv;
                }
                const uint32_t below = (1u << (8u * k)) - 1u;
                const uint32_t sumg = __builtin_amdgcn_sad_u8(a4 & below, 0u, k + 1u);  // their zeros + their ones + this one
                const uint32_t za = (a4 >> (8u * k)) & 0xFFu, zb = (b4 >> (8u * k)) & 0xFFu;
                uint32_t fz = fa < fb ? fa : fb;                                        // the zeros in front a match would take along
                fz = fz < BP_BACK ? fz : BP_BACK;
                const int score = (int)(sumg + (za < zb ? za : zb) + fz);
                const bool up = have & same & (score > best);   // (selects, not a branch)
                best = up ? score : best;
                bjq = up ? jq : bjq;
                bk = up ? k : bk;
                if (!CHAIN) break;
                jc1 = have ? nextc : 0u;
            }
            // ---- the chosen one, exactly (positions from P): agreement may continue past the third gap (periodic
            //      planes; rare), then the costs decide between this match and literals
            const bool got = best >= 0;
            if (__builtin_amdgcn_ballot_w64(got) != 0ull) {
                const uint32_t jq = got ? bjq : 1u;
                const uint32_t cc1 = P[jq], cp1 = P[jq - 1u];
                uint32_t a = jj + bk, b = jq + bk;          // P-indices of the ones the first open comparison starts at
                uint32_t pa = P[a], pb = P[b];
                uint32_t clen = pa - q1, costR = 0, tailz = 0, zl = 0;
#pragma unroll
                for (uint32_t s = 0; s < BP_PICK; ++s) {
                    const uint32_t g = (a4 >> (8u * s)) & 0xFFu;
                    costR += s < bk ? 1u + (g >= BP_TMIN + 1u ? 4u : g) : 0u;
                }
                bool act = got;
                for (uint32_t s = bk;; ++s) {   // (per lane: s starts at the lane's own bk; the trip count is what the wave needs)
                    if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
                    // (branch-free body: lanes that are done keep their values through selects; a + 1, b + 1 <= m + 4 stay inside P)
                    const uint32_t na = P[a + 1u], nbn = P[b + 1u];
                    const uint32_t ga = na - pa - 1u, gb = nbn - pb - 1u;
                    const bool cdiff = EXC && (((bp_bits(S.cls, a + 1u) ^ bp_bits(S.cls, b + 1u)) & 1u) != 0u);
                    const bool stop = (s < BP_PICK) | (ga != gb) | (ga >= 255u) | (a >= m) | (s >= BP_STEPS) | cdiff;
                    const uint32_t z = ga < gb ? ga : gb;
                    costR += act ? 1u + (ga >= BP_TMIN + 1u ? 4u : ga) : 0u;
                    clen += act ? 1u + (stop ? z : ga) : 0u;
                    tailz = (act & stop) ? ga - z : tailz;
                    zl = (act & stop) ? z : zl;
                    act = act & !stop;
                    a += act ? 1u : 0u;
                    b += act ? 1u : 0u;
                    pa = act ? na : pa;
                    pb = act ? nbn : pb;
                }
                {
                    const uint32_t gc = cc1 - cp1 - 1u;
                    uint32_t cnb = gq < gc ? gq : gc;
                    cnb = cnb < BP_BACK ? cnb : BP_BACK;
                    const uint32_t costH = 3u + (clen + cnb >= 19u ? 1u : 0u) - cnb + (tailz >= BP_TMIN ? 3u : tailz);
                    uint32_t end = (uint32_t)q + clen;
                    clamped = end > BP_MATCHLIMIT;
                    end = end < BP_MATCHLIMIT ? end : BP_MATCHLIMIT;
                    // signed compares: costH may go below zero when many zeros are pulled in
                    const int gain = (int)costR - (int)costH;
                    hv = got & (gain > 0) & ((int)end - q >= 4) & ((int)end - (q - (int)cnb) >= BP_MINM) & (q <= BP_MFLIMIT);   // (& not &&: no nest of exec-mask regions)
                    len = hv ? end - (uint32_t)q : 0u;
                    nb = hv ? cnb : 0u;
                    c1 = hv ? cc1 : 0u;
                    a_last = a;
                    z_last = zl;
                    clamped = clamped & hv;
                }
            }
        }
        BP_MARK("cand_done");
        if (LAZY) {
            // ---- one-step lazy rule (the file-writing paths' effort level; tools/sim/gapenc_ref.c states it): a one gives up
            //      its match when the NEXT one lies inside that match and has a match of its own that ends further by at least
            //      BP_LAZY_MIN + what giving up costs (the zeros this one's match was pulled back over, the zeros between this
            //      one and the start of the next one's match).  The next one is the neighbouring lane (the window's last lane
            //      keeps its match); in a run of lanes that all would give up, every other one does, counted from the far end.
            const uint32_t E0 = hv ? (uint32_t)q + len : (uint32_t)(q + 1);
            const uint32_t hvN = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(hv ? 1u : 0u), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            const uint32_t EN = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)E0, 0x130, 0xf, 0xf, false);
            const uint32_t nbN = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nb, 0x130, 0xf, 0xf, false);
            const int msN = (int)qn1 - 1 - (int)nbN;                      // where the next one's match starts
            const int between = msN - (int)q1 > 0 ? msN - (int)q1 : 0;     // zeros between this one and that start
            const bool Ln = hv & (hvN != 0u) & (lane < 63u) & (j + 1 < (int)m) & (qn1 <= E0) & ((int)EN - (int)E0 >= (int)(BP_LAZY_MIN + nb) + between);
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(Ln);
            const unsigned long long rest = ~(lm >> lane);                 // first zero bit = end of this lane's run
            const uint32_t tlo = (uint32_t)__builtin_ctz((uint32_t)rest | 0x80000000u), thi = (uint32_t)__builtin_ctz((uint32_t)(rest >> 32) | 0x80000000u);
            const uint32_t t = (uint32_t)rest != 0u ? tlo : 32u + thi;     // (lm >> lane has zeros from bit 64 - lane on: the run ends inside)
            const bool give = Ln && (t & 1u);
            hv = hv && !give;
            len = give ? 0u : len;
            nb = give ? 0u : nb;
            c1 = give ? 0u : c1;
            a_last = give ? jj : a_last;
            z_last = give ? 0u : z_last;
            clamped = clamped && !give;
        }
        const uint32_t E = hv ? (uint32_t)q + len : (uint32_t)(q + 1);   // end of what this one codes (0 for the virtual one)
        // first one at or behind E = the number of ones in front of position E.  A one that codes itself alone (E = q + 1) has
        // itself and its predecessors in front: its own P-index.  A match ends in the zeros behind the last one it covers (or
        // exactly at the next one): that one's P-index.  Only a match that was cut at the stream's match limit (a handful of
        // ones at the very end of a plane) needs the count from the bit map.
        uint32_t nxt = hv ? a_last : jj;
        if (__builtin_amdgcn_ballot_w64(clamped) != 0ull) {   // (wave-uniform, rare)
            const uint32_t w = E >> 6, bb = E & 63u, wc = w < 63u ? w : 63u;
            const uint32_t lo = bm[2u * wc], hi = bm[2u * wc + 1u];
            const uint32_t mlo = bb >= 32u ? 0xFFFFFFFFu : ((1u << bb) - 1u);
            const uint32_t mhi = bb > 32u ? ((1u << (bb - 32u)) - 1u) : 0u;
            const uint32_t in_map = (uint32_t)wpre[wc] + (uint32_t)__popc(lo & mlo) + (uint32_t)__popc(hi & mhi);
            nxt = clamped ? (w >= 64u ? m : in_map) : nxt;
        }
        BP_MARK("nxt_done");
        // ---- which ones of the window are coded: follow nxt from `cur` by pointer doubling.  (Measured against the plain
        //      walk on the scalar unit — one v_readlane + 5 scalar instructions per coded one, half the vector instructions:
        //      14.8 against 13.0 ms per step.  A wave's dependent chain is what counts, not its instruction total.)
        bool sel = false;
        if (BP_SKIP >= 3) {
            sink += nxt + E;
            continue;
        }
        if (cur < jw + 64) {   // (wave-uniform) otherwise the whole window lies inside an earlier match
            const uint32_t e0 = (uint32_t)(cur - jw);
            uint32_t jump = valid ? (nxt - (uint32_t)jw < 64u ? nxt - (uint32_t)jw : 64u) : 64u;   // nxt > j: always forward
#if BP_SELECT_SCALAR
            // the chain walked on the scalar side — one v_readlane and a handful of scalar instructions per coded one — instead
            // of six rounds of flag write -> flag read -> ds_bpermute (measured slower, see BP_SELECT_SCALAR)
            unsigned long long SEL = 0ull;
            uint32_t c = e0, last = e0;
            do {
                SEL |= 1ull << c;
                last = c;
                c = (uint32_t)__builtin_amdgcn_readlane((int)jump, (int)c);
            } while (c < 64u);
            sel = __builtin_amdgcn_inverse_ballot_w64(SEL) && valid;
            // the coded one whose successor lies outside the window hands over to the next window
            if ((int)(jw + (int)last) < (int)m) cur = (int)__builtin_amdgcn_readlane((int)nxt, (int)last);
#else
            const uint32_t p0 = jump;
            flag[lane] = lane == e0 ? 1u : 0u;
            flag[lane == 0u ? 64u : 69u] = 0u;
            BP_FENCE();
            bool reach = lane == e0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                flag[reach ? jump : 68u] = 1u;   // (no exec-mask region: a scalar instruction costs this kernel four times a vector one)
                BP_FENCE();
                reach = flag[lane] != 0u;
                const uint32_t j2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((jump & 63u) << 2), (int)jump);
                jump = jump < 64u ? j2 : 64u;
            }
            sel = reach && valid;
            const unsigned long long ex = __builtin_amdgcn_ballot_w64(sel && p0 == 64u);
            // the coded one whose successor lies outside the window hands over to the next window
            if (ex != 0ull) cur = (int)__builtin_amdgcn_readlane((int)nxt, (int)(__ffsll((long long)ex) - 1));
#endif
        }
        BP_MARK("sel_done");
        if (BP_SKIP >= 2) {
            sink += sel ? nxt : 0u;
            continue;
        }
        // ---- the coded ones that emit something — a match M (if hv) and / or a tail run T over the zeros up to the next
        //      coded one — are queued; layout and emission run on 64 of them at a time (bp_emit_batch), every lane busy,
        //      where the first version laid out and emitted per window with the ~30 % of its lanes that were coded ones
        const bool onM = sel && hv;
        // the run behind E starts one later when byte E - 1 is a one (an offset-1 run copies its predecessor): always for a one
        // that codes itself alone (and the virtual one in front of the stream), for a match when it ends with its last one
        uint32_t rs = E + ((!hv || z_last == 0u) ? 1u : 0u);
        if (__builtin_amdgcn_ballot_w64(clamped) != 0ull)   // (wave-uniform, rare: a match cut at the match limit — ask the bit map)
            rs = clamped ? E + ((bp_bits(bm, E - 1u) & 1u) ? 1u : 0u) : rs;
        uint32_t re = (uint32_t)P[(nxt < m ? nxt : m) + 1u] - 1u;   // position of the next one (n behind the last)
        re = re < BP_MATCHLIMIT ? re : BP_MATCHLIMIT;
        const bool onT = sel && (int)re - (int)rs >= BP_TMIN && rs <= BP_MFLIMIT;   // (long enough; emitted or not: bp_emit_batch)
        const bool want = onM || onT;
        const unsigned long long wm = __builtin_amdgcn_ballot_w64(want);
        const uint32_t nw = (uint32_t)__builtin_popcountll(wm);
        const bool last = jw + 64 >= (int)m;
        // one call site for the batch code: first make room if this window's coded ones would not fit (rare), then queue
        // them, then drain whole batches (everything at the end)
        for (int phase = 0; phase < 2; ++phase) {
            if (phase == 1) {
                const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
                S.queue[want ? slot : (uint32_t)BP_QCAP] = bp_entry(E, (uint32_t)(q - (int)nb), onM, onT, rs, re, q1 - c1);
                qn += nw;
                BP_MARK("queued");
            }
            // a batch leaves the queue when the entry behind its last one is there too (the run rule looks at it) — or when
            // the stream ends
            while (phase == 0 ? qn + nw > BP_QCAP : (qn >= 65u || (last && qn != 0u))) {   // (wave-uniform)
                const uint32_t keep = phase == 1 && last ? 0u : 1u;
                const uint32_t nb_ = qn - keep < 64u ? qn - keep : 64u;
                BP_FENCE();
                const BpOut r = bp_emit_batch<EXC>(S, bm, out, nb_, qn, BpOut{prev_end, gop, sop}, lane);
                prev_end = r.prev_end, gop = r.gop, sop = r.sop;
                BP_FENCE();
                const bool mvp = lane + nb_ < qn;
                const uint2 mv = S.queue[mvp ? nb_ + lane : (uint32_t)BP_QCAP];
                BP_FENCE();
                S.queue[mvp ? lane : (uint32_t)BP_QCAP] = mv;
                qn -= nb_;
            }
        }
        BP_MARK("emit_done");
    }
    BP_MARK("loop_done");
    if (BP_SKIP >= 1) {
        if (lane == 0) csize[sidx] = (prev_end + sink) & 0xFFFu;
        return;
    }
    // ---- last literals
    {
        const uint32_t ll = BP_N - prev_end, llx = bp_len_ext(ll);
        BP_FENCE();
        if (sop + 1u + llx + ll > BpCfg<EXC>::STAGE) {
            bp_flush(S.stage, out + gop, sop, lane);
            gop += sop;
            sop = 0;
        }
        const bool direct = 1u + llx + ll > BpCfg<EXC>::STAGE;
        auto tail = [&](auto dstp) {
            if (lane == 0) {
                dstp[0] = (uint8_t)((ll < 15u ? ll : 15u) << 4);
                uint32_t r = ll - 15u;
                for (uint32_t k = 0; k < llx; ++k) {
                    dstp[1u + k] = (uint8_t)(k + 1u == llx ? r : 255u);
                    r -= 255u;
                }
            }
            for (uint32_t k = lane; k < ll; k += 64u)
                dstp[1u + llx + k] = (uint8_t)bp_lit<EXC>(bp_bits(bm, prev_end + k), EXC ? bp_bits(S.xm, prev_end + k) : 0u, 0u);
        };
        if (direct) {
            tail(out + gop);
            gop += 1u + llx + ll;
        } else {
            tail(S.stage + sop);
            sop += 1u + llx + ll;
        }
    }
    uint32_t op = gop + sop;
    if (op >= BP_N) {   // incompressible: Blosc stores the (shuffled) stream verbatim
        for (uint32_t k = lane; k < BP_N; k += 64u) out[k] = (uint8_t)bp_lit<EXC>(bp_bits(bm, k), EXC ? bp_bits(S.xm, k) : 0u, 0u);
        op = BP_N;
    } else {
        BP_FENCE();
        bp_flush(S.stage, out + gop, sop, lane);
    }
    if (BP_SKIP) op = (op & 0xFFFu) + (sink & 1u);
#ifdef BP_PROBE_HANDOFF
    // development (round 4): the least a wave would pay to hand its stream to another workgroup INSIDE this launch (framing
    // fused by a look-back, or "the chunk's last stream frames the chunk"): its stores drained, then one returning
    // agent-scope atomic on a word the 256 streams of its chunk share (a spare word behind the chunk's first scratch slot).
    // LZ4 stage with it: tools/dev/lz4_handoff.sh, DESIGN.md 3.3.
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(reinterpret_cast<uint32_t *>(scratch + (sidx & ~255ull) * slot_bytes + slot_bytes - 8u), 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        op += old == 0xFFFFFFF0u ? 1u : 0u;   // (keeps the returned value alive; never true)
    }
#endif
    if (lane == 0) csize[sidx] = op;
}

// grid = number of 8 KiB blocks; 128 threads: wave w codes byte plane w of the block
// DEPTH = candidates tried per one along the hash chain (1: the table's entry only; 0: no hash matches at all,
// offset-1 runs only — the fastest level)
// PLANES: src is the bit-plane form of the matrix (include/hhgt.h, tile-major: common.h; written by k_encode_planes): the
// wave's bit map is 16 pieces of 32 bytes it gathers as they are (one load instruction: lane l takes 8 bytes of tile
// l / 4), and a set EXC bit is what "a byte > 1" was.  The pieces of four neighbouring sample rows share a 128-byte
// line, so the blocks are dealt to the workgroups in an XCD-aware order: of 64 consecutive workgroups the eight that land
// on one XCD (round robin) take eight consecutive blocks — one L2 fetches each line once.
// EXC (with PLANES): the exception-aware instantiation — planes whose bytes are 0, 1 or 0xF7 (missing calls; config 4).
// On a fixed grid every wave scans the stream sizes of its plane, 64 blocks per load, and codes the streams the plain
// instantiation left marked (csize = 0xFFFFFFFF; a list of marked blocks built by atomic adds on one counter cost more than
// the coding itself when every plane is marked: 6 ms for config 4's 537 k blocks); the bit map is the map of NONZERO bytes, a second map says which of them are 0xF7, every one
// carries that bit as its class, and two ones only agree if their classes do (tools/sim/gapenc_ref.c states the rules).
// Streams it cannot code either (a call beyond 0 / 1 / missing, too many nonzero bytes) stay marked for the byte-wise kernel.
template <int DEPTH, bool PLANES, bool EXC, bool LAZY = false>
__global__ __launch_bounds__(128, EXC ? BP_WAVES_EXC : BP_WAVES) void k_lz4_bitplanes(const uint8_t *__restrict__ src, PlanesGeom pg, uint32_t n_blocks,
                                                          uint8_t *__restrict__ scratch, uint64_t slot_bytes, uint32_t *__restrict__ csize,
                                                          uint32_t *__restrict__ flags, uint32_t tag)
{
    static_assert(!EXC || PLANES, "the exception-aware coder reads bit planes");
    // flags (may be null): flags[0] == tag <=> the plain instantiation of THIS call left a stream marked, flags[1] == tag <=> the
    // exception-aware one did.  The scanning launches behind (this kernel with EXC, lz4.hip's kernel in scan mode) look at the
    // word first and are gone if nobody marked anything — on the headline cohort that is every call, and the two scans were
    // 48 us of each shard's chain (r04b_bench_kernel_stats.csv).  A tag per call instead of a flag somebody has to clear.
    if (EXC && flags && __builtin_nontemporal_load(flags) != tag) return;
    constexpr bool CHAIN = DEPTH > 1;
    __shared__ BpLds<CHAIN, EXC> lds[2];
    __shared__ uint32_t nonbin[2][2];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    // (EXC: the two waves of a workgroup are independent; each walks the marked streams of its plane)
    for (uint32_t b0 = EXC ? blockIdx.x * 64u : 0u; b0 < (EXC ? n_blocks : 1u); b0 += EXC ? gridDim.x * 64u : 1u) {
    unsigned long long todo = 1ull;
    if (EXC) {   // which of the 64 blocks from b0 on still have this wave's stream marked
        const uint32_t bb = b0 + lane;
        todo = __builtin_amdgcn_ballot_w64(bb < n_blocks && csize[(uint64_t)bb * 2u + wave] == 0xFFFFFFFFu);
    }
    while (todo != 0ull) {   // (wave-uniform)
    {
    const uint32_t bid = EXC ? b0 + (uint32_t)__builtin_ctzll(todo)
                             : PLANES ? ((blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u)) : blockIdx.x;
    todo &= todo - 1ull;
    if (PLANES && !EXC && bid >= n_blocks) return;   // (the grid is rounded up to whole groups of 64)
    uint32_t xlo = 0, xhi = 0;   // EXC: the lane's 64 positions of the missing-call map
    uint32_t wlo, whi;
    bool nonbinary;
    if (PLANES) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        uint64_t col;
        uint32_t row, bi;
        planes_block(pg, bid, &col, &row, &bi);
        const u32x2 *pl = reinterpret_cast<const u32x2 *>(src + planes_piece(pg, col, bi * 16u + (lane >> 2), wave, row)) + (lane & 3u);
        const u32x2 one = *pl, exc = *(pl + (uint64_t)pg.S_pad * 8ull);   // EXC: two kind-planes (S_pad * 32 bytes each) further
        wlo = one.x;
        whi = one.y;
        lds[wave].bm[2u * lane] = wlo;
        lds[wave].bm[2u * lane + 1u] = whi;
        if (lane < 4u) lds[wave].bm[128u + lane] = 0u;
        if (EXC) {
            // (ONE, EXC) = (0, 1): a call beyond 0 / 1 / missing, its byte lives in the int8 matrix — not this coder's
            nonbinary = __builtin_amdgcn_ballot_w64(((exc.x & ~one.x) | (exc.y & ~one.y)) != 0u) != 0ull;
            xlo = exc.x;
            xhi = exc.y;
            lds[wave].xm[2u * lane] = xlo;
            lds[wave].xm[2u * lane + 1u] = xhi;
            if (lane < 4u) lds[wave].xm[128u + lane] = 0u;
            if (lane < 24u) lds[wave].cls[lane] = 0u;
        } else {
            nonbinary = __builtin_amdgcn_ballot_w64((exc.x | exc.y) != 0u) != 0ull;
        }
    } else {
    const uint8_t *blk = src + (uint64_t)bid * 8192u;

    // ---- phase A: wave r packs the bits of block bytes [4096 r, 4096 r + 4096) for BOTH planes (byte-shuffle fused:
    //      even bytes are plane 0, odd bytes plane 1).  Load k of a lane is the 16 bytes at 1024 k + 16 lane: every load
    //      instruction of the wave covers one contiguous KiB (the first version gave each lane 64 contiguous bytes, i.e.
    //      four instructions that each touched all 64 lines of the half block; FETCH_SIZE is the same either way — the L2
    //      absorbed the repeats — but the step is 1.5 % faster with this form).
    //      16 bytes = 8 positions of each plane = one BYTE of each bit map, at byte 256 r + 64 k + lane.
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *p = reinterpret_cast<const u32x4 *>(blk + 4096u * wave + 16u * lane);
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);   // streamed once
        uint32_t orall = 0;
        uint8_t *bm0 = reinterpret_cast<uint8_t *>(lds[0].bm) + 256u * wave + lane;
        uint8_t *bm1 = reinterpret_cast<uint8_t *>(lds[1].bm) + 256u * wave + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t x = d[i];
                orall |= x;
                const uint32_t xm = x & 0x01010101u;   // (a byte > 1 in one plane must not leak into the other plane's bits)
                const uint32_t y = xm | (xm >> 15);    // bits 0,1: plane-0 bytes; bits 8,9: plane-1 bytes
                acc0 |= (y & 3u) << (2 * i);
                acc1 |= ((y >> 8) & 3u) << (2 * i);
            }
            bm0[64 * k] = (uint8_t)acc0;
            bm1[64 * k] = (uint8_t)acc1;
        }
        const uint32_t nb0 = orall & 0x00FE00FEu, nb1 = orall & 0xFE00FE00u;   // a byte > 1 somewhere in plane 0 / plane 1
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(nb0 != 0u), b1 = __builtin_amdgcn_ballot_w64(nb1 != 0u);
        if (lane == 0) {
            nonbin[wave][0] = b0 != 0ull;
            nonbin[wave][1] = b1 != 0ull;
        }
        if (lane < 4u) lds[wave].bm[128u + lane] = 0u;
    }
    __syncthreads();
    wlo = lds[wave].bm[2u * lane];
    whi = lds[wave].bm[2u * lane + 1u];
    nonbinary = (nonbin[0][wave] | nonbin[1][wave]) != 0u;
    }
    BP_MARK("A_done");
    if (BP_SKIP >= 5) {
        if (lane == 0) csize[(uint64_t)bid * 2u + wave] = lds[wave].bm[5] & 1u;
        continue;
    }

    bp_code_plane<DEPTH, EXC, LAZY, EXC ? 1 : 0, !EXC>(lds[wave], wlo, whi, xlo, xhi, nonbinary, bid, wave, lane, scratch, slot_bytes, csize, flags, tag);
    }
    }
    }
}

// The plain and the exception-aware coder in ONE launch over bit planes (round 4): a wave looks at its plane's missing-call map
// and takes the one or the other path (wave-uniform; both are the instantiations above, inlined).  What it replaces is the plain
// launch marking every plane that holds a missing call for a second, scanning launch — on config 4, where every plane does,
// 0.4 ms of reading all planes for nothing.  Planes neither can code (a call beyond 0 / 1 / missing, too many nonzero bytes) are
// marked for the byte-wise kernel as before (flags[0]).
template <int DEPTH, bool LAZY>
__global__ __launch_bounds__(128, BP_WAVES) void k_lz4_bitplanes_uni(const uint8_t *__restrict__ src, PlanesGeom pg, uint32_t n_blocks,
                                                                      uint8_t *__restrict__ scratch, uint64_t slot_bytes,
                                                                      uint32_t *__restrict__ csize, uint32_t *__restrict__ flags, uint32_t tag)
{
    constexpr bool CHAIN = DEPTH > 1;
    union BpBoth {
        BpLds<CHAIN, false> p;
        BpLds<CHAIN, true> x;
    };
    __shared__ BpBoth lds[2];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t bid = (blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u);   // (XCD-aware order, as above)
    if (bid >= n_blocks) return;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint64_t col;
    uint32_t row, bi;
    planes_block(pg, bid, &col, &row, &bi);
    const u32x2 *pl = reinterpret_cast<const u32x2 *>(src + planes_piece(pg, col, bi * 16u + (lane >> 2), wave, row)) + (lane & 3u);
    const u32x2 one = *pl, exc = *(pl + (uint64_t)pg.S_pad * 8ull);
    if (__builtin_amdgcn_ballot_w64((exc.x | exc.y) != 0u) == 0ull) {   // no missing call in this plane: the plain coder
        BpLds<CHAIN, false> &S = lds[wave].p;
        S.bm[2u * lane] = one.x;
        S.bm[2u * lane + 1u] = one.y;
        if (lane < 4u) S.bm[128u + lane] = 0u;
        bp_code_plane<DEPTH, false, LAZY, 0, true>(S, one.x, one.y, 0u, 0u, false, bid, wave, lane, scratch, slot_bytes, csize, flags, tag);
    } else {
        BpLds<CHAIN, true> &S = lds[wave].x;
        // (ONE, EXC) = (0, 1): a call beyond 0 / 1 / missing, its byte lives in the int8 matrix — not this coder's
        const bool nonbinary = __builtin_amdgcn_ballot_w64(((exc.x & ~one.x) | (exc.y & ~one.y)) != 0u) != 0ull;
        S.bm[2u * lane] = one.x;
        S.bm[2u * lane + 1u] = one.y;
        if (lane < 4u) S.bm[128u + lane] = 0u;
        S.xm[2u * lane] = exc.x;
        S.xm[2u * lane + 1u] = exc.y;
        if (lane < 4u) S.xm[128u + lane] = 0u;
        if (lane < 24u) S.cls[lane] = 0u;
        bp_code_plane<DEPTH, true, LAZY, 0, true>(S, one.x, one.y, exc.x, exc.y, nonbinary, bid, wave, lane, scratch, slot_bytes, csize, flags, tag);
    }
}

// depth: candidates per one; + 0x100: with the lazy rule (instantiated for 12 candidates — clevel 9 — and, for measurements, 2)
int launch_lz4_bitplanes(const uint8_t *d_src, bool planes, PlanesGeom pg, uint64_t n_blocks, uint8_t *d_scratch, size_t slot_bytes, uint32_t *d_csize,
                         int depth, uint32_t *d_flags, uint32_t tag, bool *exc_ran, hipStream_t st)
{
    if (exc_ran) *exc_ran = false;
    const bool lazy = (depth & 0x100) != 0;
    depth &= 0xFF;
    if (n_blocks == 0) return HHGT_OK;
    if (n_blocks > 0x7fffffffull) {
        hhgt_set_error("lz4: too many blocks");
        return HHGT_ERR_ARG;
    }
    // development: extra (unused) dynamic LDS per workgroup caps how many workgroups a CU holds, which leaves LDS and
    // wave slots to a kernel running beside this one on another stream
    static const int lds_pad = getenv("HHGT_LZ4_LDS_PAD") ? atoi(getenv("HHGT_LZ4_LDS_PAD")) : 0;
    // the exception-aware instantiation scans for the streams the plain one marked, on a grid that fills the chip
    // (HHGT_LZ4_EXC=0: every marked stream goes to the byte-wise kernel, as before round 3)
    static const bool exc_env = !(getenv("HHGT_LZ4_EXC") && atoi(getenv("HHGT_LZ4_EXC")) == 0);
    // planes: one launch in which every wave takes the plain or the exception-aware path by its plane's missing-call map
    // (HHGT_LZ4_UNI=0: the plain launch marks, the exception-aware one scans for the marks — rounds 3 and 4a-c)
    static const bool uni_env = exc_env && !(getenv("HHGT_LZ4_UNI") && atoi(getenv("HHGT_LZ4_UNI")) == 0);
    const uint32_t exc_grid = (uint32_t)((n_blocks + 63) / 64 < 256u * 14u ? (n_blocks + 63) / 64 : 256u * 14u);
#define BP_LAUNCH2(D, PL, LZ)                                                                                               \
    hipLaunchKernelGGL((k_lz4_bitplanes<D, PL, false, LZ>), dim3(PL ? (uint32_t)((n_blocks + 63) / 64 * 64) : (uint32_t)n_blocks), dim3(128), lds_pad, st, \
                       d_src, pg, (uint32_t)n_blocks, d_scratch, (uint64_t)slot_bytes, d_csize, d_flags, tag)
#define BP_LAUNCH(D, LZ)                                                                                                    \
    do {                                                                                                                    \
        if (planes && uni_env) {                                                                                            \
            hipLaunchKernelGGL((k_lz4_bitplanes_uni<D, LZ>), dim3((uint32_t)((n_blocks + 63) / 64 * 64)), dim3(128), lds_pad, st, d_src, pg, \
                               (uint32_t)n_blocks, d_scratch, (uint64_t)slot_bytes, d_csize, d_flags, tag);                 \
        } else if (planes) {                                                                                                \
            BP_LAUNCH2(D, true, LZ);                                                                                        \
            if (exc_env) {                                                                                                  \
                hipLaunchKernelGGL((k_lz4_bitplanes<D, true, true, LZ>), dim3(exc_grid), dim3(128), 0, st, d_src, pg, (uint32_t)n_blocks, d_scratch, \
                                   (uint64_t)slot_bytes, d_csize, d_flags, tag);                                            \
                if (exc_ran) *exc_ran = true;                                                                               \
            }                                                                                                               \
        } else                                                                                                              \
            BP_LAUNCH2(D, false, LZ);                                                                                       \
    } while (0)
    if (lazy && depth == 2) BP_LAUNCH(2, true);
    else if (lazy) BP_LAUNCH(12, true);
    else if (depth <= 0) BP_LAUNCH(0, false);
    else if (depth == 1) BP_LAUNCH(1, false);
    else if (depth == 2) BP_LAUNCH(2, false);
    else if (depth <= 4) BP_LAUNCH(4, false);
    else if (depth <= 8) BP_LAUNCH(8, false);
    else if (depth <= 12) BP_LAUNCH(12, false);
    else BP_LAUNCH(16, false);
#undef BP_LAUNCH
#undef BP_LAUNCH2
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
