// encode.hip — GT text -> int8 genotype tiles on gfx950 (CDNA4).
//
// Replaces the reference's per-record, per-sample loop
//   var.getGenotypes(gt); phase1 = (int8)gt[0]; phase2 = (int8)gt[1]   /root/reference/cpp/parse_vcf.cpp:45-52
//   '.' -> -9, else allele index                                        /root/reference/cpp/vcfpp.h:546-588
// (and htslib's vcf_parse_format GT tokeniser behind it) by ONE pass over the sample columns of all
// kept lines that emits every sample at once as G[s, v', 2].
//
// k_encode_tiles<TV> (fixed-width lines "a|b\t" x S — the 1000G shape): a 256-thread workgroup owns a
//   TV-variant x 256-sample tile (TV = 64 by default: 32 KiB of LDS, four workgroups per CU).  Wave w reads
//   TV/4 lines; lane l reads 16 contiguous bytes (4 samples) of each line, so every wave-load is a coalesced
//   1 KiB run of raw GT bytes.  Each lane packs its 4 samples x TV/4 variants in registers, the tile is
//   transposed through an XOR-swizzled LDS image (conflict-free ds_write_b128 / ds_read_b128) and leaves as
//   2*TV-byte sample-row segments.
//   HBM roofline: algorithmic bytes per variant = 4*S read + 2*S written; no MFMA (byte work).
// k_encode_general (anything else: GT:DP columns, multi-digit alleles, haploid calls ...): one wave
//   per line, tab-rank by ballot/prefix over 1 KiB pieces, htslib GT rule per field.
#include "common.h"
#include <stdlib.h>

// -------------------------------------------------------------------------------------------------
// one "a|b\t" field in a dword (little endian: a, sep, b, terminator) -> h0 | h1 << 8 ; bit 31 = not
// a fixed-width field this path may decode
__device__ __forceinline__ uint32_t parse_field_exact(uint32_t x)
{
    uint32_t a = x & 0xFFu, sep = (x >> 8) & 0xFFu, b = (x >> 16) & 0xFFu, t = x >> 24;
    uint32_t da = a - '0', db = b - '0';
    bool ok = (sep == '|' || sep == '/') && t == '\t' && (da < 10u || a == '.') && (db < 10u || b == '.');
    uint32_t h0 = a == '.' ? 0xF7u : da;  // -9 as int8
    uint32_t h1 = b == '.' ? 0xF7u : db;
    return (h0 & 0xFFu) | ((h1 & 0xFFu) << 8) | (ok ? 0u : 0x80000000u);
}

__device__ __forceinline__ uint32_t parse_field(uint32_t x)
{
    // common case: both alleles are '0' or '1', separator '|', terminator '\t'
    if ((x & 0xFFFEFFFEu) == 0x09307C30u) return (x & 1u) | ((x >> 8) & 0x100u);
    return parse_field_exact(x);
}

typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4_al __attribute__((ext_vector_type(4)));

#define FILL_FIELD 0x09307C30u  // "0|0\t"
#ifndef TILE_G
#define TILE_G 16  // lines whose 16-byte loads are in flight together per wave (1 KiB each)
#endif

// TV = variants per tile (128 or 64).  LDS = 256 rows x 2*TV bytes (64 / 32 KiB): TV = 64 lets four
// workgroups share a CU, so the load, transpose and store phases of different tiles overlap.
template <int TV>
__global__ __launch_bounds__(256, TV == 128 ? 2 : 4) void k_encode_tiles(
    const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ k_soff,
    const uint32_t *__restrict__ k_meta, const uint64_t *__restrict__ d_cursor, LayoutDev lay, int8_t *__restrict__ G,
    uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag, DevCounters *cnt)
{
    const uint64_t v_base = *d_cursor;   // append position: device-resident, so a chain of calls needs no host round trip
    constexpr int LW = TV / 4;        // lines per wave
    constexpr int SLOTS = TV / 8;     // 16-byte slots per LDS row
    constexpr int WSL = SLOTS / 4;    // slots per row owned by one wave
    constexpr int G_ = LW < TILE_G ? LW : TILE_G;
    __shared__ uint4 tile[TILE_S * SLOTS];  // 256 rows x 2*TV B, 16-byte slots XOR-swizzled by (row >> 2) & 7
    const uint32_t n_kept = (uint32_t)cnt->n_kept;
    const uint64_t gv0 = (v_base / TV + blockIdx.x) * (uint64_t)TV;  // first global column of tile
    const long long k0 = (long long)gv0 - (long long)v_base;         // batch-local kept index of it
    if (k0 >= (long long)n_kept || (!lay.ring && gv0 >= lay.v_capacity)) return;
    const uint32_t s0 = blockIdx.y * TILE_S;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t S = lay.S;
    const uint32_t ls = s0 + 4u * lane;                                 // first sample of this lane
    const uint32_t nval = ls >= S ? 0u : (S - ls >= 4u ? 4u : S - ls);  // samples this lane owns
    const uint32_t last_q = (S - 1u >= ls && S - 1u < ls + 4u) ? S - 1u - ls : 4u;  // which dword is sample S-1

    uint32_t acc[4][LW / 2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < LW / 2; ++c) acc[q][c] = 0;

#pragma unroll
    for (int g = 0; g < LW / G_; ++g) {
        uint4 raw[G_];
        bool lvalid[G_];
#pragma unroll
        for (int j = 0; j < G_; ++j) {
            const long long k = k0 + (long long)(w * LW + g * G_ + j);
            bool valid = k >= 0 && k < (long long)n_kept;
            uint32_t meta = 0, soff = 0;
            if (valid) {
                meta = k_meta[k];
                soff = k_soff[k];
            }
            valid = valid && (meta & LF_FAST);
            lvalid[j] = valid;
            uint4 v = make_uint4(FILL_FIELD, FILL_FIELD, FILL_FIELD, FILL_FIELD);
            if (valid && nval) {
                const uint64_t off = (uint64_t)soff + 4ull * ls;
                if (nval == 4u && off + 16ull <= n) {
                    u32x4_unaligned t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(text + off));
                    v = make_uint4(t.x, t.y, t.z, t.w);
                } else {
                    uint32_t d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t x = FILL_FIELD;
                        if ((uint32_t)q < nval) {
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                uint64_t idx = off + (uint64_t)(q * 4 + bb);
                                uint32_t c = idx < n ? text[idx] : (uint32_t)'\t';
                                x = (x & ~(0xFFu << (bb * 8))) | (c << (bb * 8));
                            }
                        }
                        d[q] = x;
                    }
                    v = make_uint4(d[0], d[1], d[2], d[3]);
                }
            }
            raw[j] = v;
        }
#pragma unroll
        for (int j = 0; j < G_; ++j) {
            uint32_t x[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
            uint32_t bad = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t xx = x[q];
                // the last sample of a line is terminated by the line end, not by a tab (the region
                // length check LF_FAST already pinned where it ends)
                if ((uint32_t)q == last_q) xx = (xx & 0x00FFFFFFu) | 0x09000000u;
                uint32_t r = parse_field(xx);
                bad |= r;
                const int col = g * G_ + j;
                acc[q][col >> 1] |= (r & 0xFFFFu) << ((col & 1) * 16);
            }
            if (lvalid[j]) {
                unsigned long long bm = __ballot((bad & 0x80000000u) != 0u);
                if (bm != 0ull && lane == 0) {
                    const uint32_t k = (uint32_t)(k0 + (long long)(w * LW + g * G_ + j));
                    if (atomicExch(&redo_flag[k], 1u) == 0u) {
                        unsigned long long slot = atomicAdd(&cnt->n_general, 1ull);
                        redo_list[slot] = k;
                    }
                }
            }
        }
    }
    // registers -> LDS (sample-major rows).  Wave w owns byte columns [w * TV/2, (w+1) * TV/2) of every row.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t row = 4u * lane + q;
        const uint32_t swz = (row >> 2) & 7u;
#pragma unroll
        for (int c = 0; c < WSL; ++c) {
            const uint32_t slot = (w * WSL + c) ^ (swz & (SLOTS - 1));
            tile[row * SLOTS + slot] = make_uint4(acc[q][c * 4 + 0], acc[q][c * 4 + 1], acc[q][c * 4 + 2], acc[q][c * 4 + 3]);
        }
    }
    __syncthreads();
    // LDS -> HBM: each row leaves as one contiguous 2*TV-byte run (SLOTS lanes x 16 B)
    const uint32_t c16 = threadIdx.x & (SLOTS - 1);
    uint64_t vcol = gv0 / lay.Vc;
    const uint64_t vin = gv0 - vcol * lay.Vc;
    if (lay.ring) vcol %= lay.ring;
    const long long kfirst = k0 + (long long)c16 * 8;  // batch-local kept index of this lane's 8 columns
    constexpr int ROWS_PER_IT = 256 / SLOTS;
#pragma unroll 4
    for (int it = 0; it < TILE_S / ROWS_PER_IT; ++it) {
        const uint32_t row = it * ROWS_PER_IT + (threadIdx.x / SLOTS);
        const uint32_t s = s0 + row;
        if (s >= S) continue;
        const uint32_t swz = (row >> 2) & 7u;
        uint4 v = tile[row * SLOTS + (c16 ^ (swz & (SLOTS - 1)))];
        const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (s >> lay.sc_log2);
        const uint32_t sin = s - scol * lay.Sc;
        int8_t *dst = G + ((((vcol * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc + vin) * 2ull) + c16 * 16u;
        if (kfirst >= 0) {
            __builtin_nontemporal_store(u32x4_al{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_al *>(dst));
        } else if (kfirst + 8 > 0) {  // tile straddles v_base: keep the columns the previous batch wrote
            uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (kfirst + e >= 0)
                    reinterpret_cast<uint16_t *>(dst)[e] = (uint16_t)(d[e >> 1] >> ((e & 1) * 16));
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_encode_planes: the same pass as k_encode_tiles for callers whose only consumer of the matrix is the compressor —
// two BITS per allele instead of a byte (include/hhgt.h "Bit-plane form"): ONE (allele 1, or missing) and EXC (anything
// but 0 / 1), so what crosses HBM between the two halves of the path is S/2 bytes per variant each way instead of 2 S.
//
// A 256-thread workgroup owns one plane tile (PL_TILE = 256 variants) of 256 samples; wave w reads lines [64 w, 64 w + 64)
// of the tile in groups of lines (1 KiB each, 16 B per lane = 4 samples) whose loads are in flight together; where a
// group's line starts come from: pl_load_step.  Packing is SWAR on the field dword "a|b\t": (x & 0x00010001) holds the
// two allele bits, shifted by the line number they accumulate into a register whose low half is haplotype 0 and whose
// high half is haplotype 1 over 16 lines; one xor/or chain per line tells whether all four fields of the lane were
// "[01]|[01]\t" (pl_pack_group: what happens otherwise).  Per 32 lines a lane holds one dword per (sample, plane, kind):
// 16 ds_write_b32 into a 32 KiB image [kind][plane][sample][32 B]; after one barrier the image leaves as four contiguous
// 8 KiB runs — the planes are TILE-MAJOR in HBM (common.h) for exactly that.  The tile that straddles the append position
// merges with the bits the previous call wrote.
// HBM roofline: algorithmic bytes per variant = 4 S read + S/2 written.  What bounds it in practice: the 16-byte loads
// start at the byte where a line's sample columns start, and a wave-load whose lanes are not dword-aligned is worth
// ~20 % less (tools/micro/strided_read.hip: 6.4 -> 5.1 TB/s).
#define PT_V 256           // = PL_TILE
#define PT_ROWDW (PT_V / 32)                    // dwords per image row
#ifndef PT_NW
#define PT_NW 4            // waves per workgroup
#endif
#define PT_LW (PT_V / PT_NW)   // lines per wave
#ifndef PT_G
#define PT_G 8             // lines per load group
#endif
#ifndef PT_DB
#define PT_DB 1            // the next group's loads are issued before a group is packed
#endif
#ifndef PT_WGS
#define PT_WGS 4           // waves per SIMD the register allocation leaves room for (= workgroups per CU)
#endif
#define PT_C 0x09307C30u   // "0|0\t"

struct PlGroup {
    uint4 raw[PT_G];
    uint32_t lvalid;   // bit j: line j of the group is a kept fixed-width line (wave-uniform)
    uint32_t lstride;  // bit j: line j is a kept record whose columns are all of one width of 5 .. 8 bytes, GT first (k_meta bits 20-23)
};

// the line table of 32 lines at once: lane j holds soff / meta of line kb + j (meta = 0 beyond the batch), so that a
// load group needs no scalar loads (the first version paid two dependent s_load round trips in front of every line)
struct PlStep {
    uint32_t soff, meta;
};

__device__ __forceinline__ PlStep pl_load_step(const uint32_t *__restrict__ k_soff, const uint32_t *__restrict__ k_meta, long long kb,
                                               uint32_t n_kept, uint32_t lane)
{
    PlStep t;
    const long long k = kb + (long long)(lane & 31u);
    const bool valid = k >= 0 && k < (long long)n_kept;
    t.soff = valid ? k_soff[k] : 0u;
    t.meta = valid ? k_meta[k] : 0u;
    return t;
}

// the 16 bytes (four sample fields) of one line that this lane owns
template <bool EDGE>
__device__ __forceinline__ uint4 pl_load_line(const uint8_t *__restrict__ text, uint64_t n, uint32_t soff, uint32_t ls, uint32_t nval,
                                              uint32_t last_q)
{
    uint4 v = make_uint4(FILL_FIELD, FILL_FIELD, FILL_FIELD, FILL_FIELD);
    const uint64_t off = (uint64_t)soff + 4ull * ls;
    if (!EDGE) {
        // Every lane owns four samples in front of the line's last one: the 16 bytes lie inside the line.  (They start
        // wherever the line's sample columns start; in isolation a wave-load whose lanes are not dword-aligned runs at ~80 % of
        // an aligned one — tools/micro/strided_read.hip.  Two aligned forms were built and measured in round 3, both correct,
        // neither faster: aligned chunks + the neighbour lane's data by wavefront shift (the 65th chunk needs its own load per
        // line), and two aligned loads per lane with a deferred funnel shift (0.61 against 0.56 ms per chr1-sized shard;
        // both in git history: round 3).)
        u32x4_unaligned t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(text + off));
        v = make_uint4(t.x, t.y, t.z, t.w);
    } else {
        if (nval == 4u && off + 16ull <= n) {
            u32x4_unaligned t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(text + off));
            v = make_uint4(t.x, t.y, t.z, t.w);
        } else if (nval) {
            uint32_t d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t x = FILL_FIELD;
                if ((uint32_t)q < nval) {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        uint64_t idx = off + (uint64_t)(q * 4 + bb);
                        uint32_t c = idx < n ? text[idx] : (uint32_t)'\t';
                        x = (x & ~(0xFFu << (bb * 8))) | (c << (bb * 8));
                    }
                }
                d[q] = x;
            }
            v = make_uint4(d[0], d[1], d[2], d[3]);
        }
        // the last sample of a line is terminated by the line end (LF_FAST pinned where it ends)
        if (last_q == 0u) v.x = (v.x & 0x00FFFFFFu) | 0x09000000u;
        if (last_q == 1u) v.y = (v.y & 0x00FFFFFFu) | 0x09000000u;
        if (last_q == 2u) v.z = (v.z & 0x00FFFFFFu) | 0x09000000u;
        if (last_q == 3u) v.w = (v.w & 0x00FFFFFFu) | 0x09000000u;
    }
    return v;
}

struct __attribute__((packed)) PlU32 {
    uint32_t v;
};

template <bool EDGE>
__device__ __forceinline__ void pl_load_group(PlGroup &gr, const uint8_t *__restrict__ text, uint64_t n, const PlStep &stp, const int j0,
                                              uint32_t ls, uint32_t nval, uint32_t last_q)
{
    uint32_t lv = 0, lst = 0;
#pragma unroll
    for (int j = 0; j < PT_G; ++j) {
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)stp.meta, j0 + j);
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readlane((int)stp.soff, j0 + j);
        uint4 v = make_uint4(FILL_FIELD, FILL_FIELD, FILL_FIELD, FILL_FIELD);
        if (meta & LF_FAST) {   // (wave-uniform)
            lv |= 1u << j;
            v = pl_load_line<EDGE>(text, n, soff, ls, nval, last_q);
        } else if ((meta >> 20) & 15u)
            lst |= 1u << j;   // decoded by the second level below, at its stride
        gr.raw[j] = v;
    }
    gr.lvalid = lv;
    gr.lstride = lst;
}

// PT_G lines -> per sample q: one[q] / exc[q], bit sh + j (haplotype 0) and bit 16 + sh + j (haplotype 1) for line j.
// Three levels.  (1) every field of every lane is "[01]|[01]\t": ~14 vector instructions per line, unrolled.  (2) some
// lane saw something else: the group again, line by line in a rolled loop that reads the (cache-hot) line once more and
// classifies every field over the alphabet {0, 1, .} x {|, /} — branch-free SWAR on the field dword, x ^ "0|0\t": an
// allele byte must be 0x00, 0x01 or 0x1E ('.'), bit 4 of it is the EXC bit and bit 0 | bit 4 the ONE bit; the separator
// byte 0x00 or 0x53 ('/'), the terminator 0x00.  (3) a field outside that alphabet (a third allele, a multi-digit index,
// anything malformed) contributes zero bits and sends its line to the variable-width kernel, which sets the bits of the
// whole line (idempotent for the fields that were fine here).  Keeping (2) rolled and out of the registers of (1) is
// what lets four workgroups share a CU.
template <bool EDGE>
__device__ __forceinline__ void pl_pack_group(const PlGroup &gr, const int sh, uint32_t (&one)[4], uint32_t (&exc)[4], long long kbase,
                                              const uint8_t *__restrict__ text, uint64_t n, const PlStep &stp, const int j0, uint32_t ls,
                                              uint32_t nval, uint32_t last_q, uint32_t *__restrict__ redo_list,
                                              uint32_t *__restrict__ redo_flag, DevCounters *cnt, uint32_t lane, uint32_t S_all)
{
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, bad = 0;
#pragma unroll
    for (int j = 0; j < PT_G; ++j) {
        const uint4 x = gr.raw[j];
        a0 |= (x.x & 0x00010001u) << (sh + j);
        a1 |= (x.y & 0x00010001u) << (sh + j);
        a2 |= (x.z & 0x00010001u) << (sh + j);
        a3 |= (x.w & 0x00010001u) << (sh + j);
        bad |= ((x.x ^ PT_C) | (x.y ^ PT_C)) | ((x.z ^ PT_C) | (x.w ^ PT_C));
    }
    bad |= gr.lstride ? 2u : 0u;   // a record decoded at its stride is the second level's (as data, not as a branch: the branch cost 86 spilled registers)
    if (__builtin_amdgcn_ballot_w64((bad & 0xFFFEFFFEu) != 0u) == 0ull) {
        one[0] |= a0, one[1] |= a1, one[2] |= a2, one[3] |= a3;
        return;
    }
#pragma unroll 1
    for (int j = 0; j < PT_G; ++j) {
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)stp.meta, j0 + j);
        const uint32_t wd = (meta >> 20) & 15u;
        if (!(meta & LF_FAST) && wd == 0u) continue;
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readlane((int)stp.soff, j0 + j);
        uint32_t hard = 0;
        // one field "a|b\t" over the alphabet {0, 1, .} x {|, /}: bits into one[q] / exc[q], anything else -> hard
#define PL_CLASSIFY(XQ, Q)                                                                                            \
    do {                                                                                                               \
        const uint32_t y = (XQ) ^ PT_C;                                                                                \
        const uint32_t t4 = y >> 4;                                                                                    \
        const uint32_t e2 = t4 & 0x00010001u;                              /* '.' in either allele */                  \
        const uint32_t o2 = (y | t4) & 0x00010001u;                        /* '1' or '.' */                            \
        const uint32_t want = __umul24(e2, 30u) | (y & 0x00010001u & ~e2); /* what the allele bytes must then be */    \
        const uint32_t z = y & 0xFF00FF00u, zz = z ^ 0x00005300u;          /* separator / terminator bytes */          \
        const uint32_t off = ((y & 0x00FF00FFu) ^ want) | (z < zz ? z : zz);                                           \
        const uint32_t keep = off ? 0u : 0xFFFFFFFFu;                                                                  \
        one[Q] |= (o2 & keep) << (sh + j);                                                                             \
        exc[Q] |= (e2 & keep) << (sh + j);                                                                             \
        hard |= off;                                                                                                   \
    } while (0)
        if (meta & LF_FAST) {   // (wave-uniform)
            const uint4 v = pl_load_line<EDGE>(text, n, soff, ls, nval, last_q);
            PL_CLASSIFY(v.x, 0);
            PL_CLASSIFY(v.y, 1);
            PL_CLASSIFY(v.z, 2);
            PL_CLASSIFY(v.w, 3);
        } else {
            // a record whose columns are all `wd` (5 .. 8) bytes wide with the GT sub-field first — "a|b:dd\t": config 4's GT:DP
            // records (round 3 sent them all to the variable-width kernel: 3.5 of config 4's 16 ms, tab ranking and one atomic
            // per nonzero call).  Each column is checked where it stands — byte 3 is the ':' that ends the GT, no tab or
            // newline up to the column's last byte, which is the tab (the newline for the last sample) — and classified as
            // "a|b\t"; a column of any other shape is a field no alphabet accepts, which sends the line to the variable-width
            // kernel like any other surprise.  One field at a time: the first level's registers stay what they were.
            const uint32_t shb = 8u * (wd - 5u);   // where the column's last byte sits in its second dword
            const uint8_t *lbase = text + soff;    // (wave-uniform base, 32-bit lane offsets: no 64-bit address pairs per lane)
            const uint64_t room64 = n - (uint64_t)soff;
            const uint32_t room = room64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)room64;
            const uint32_t lowm = shb ? (1u << shb) - 1u : 0u;
#define PL_STRIDED(Q)                                                                                                  \
    do {                                                                                                               \
        uint32_t x = FILL_FIELD;                                                                                       \
        if ((uint32_t)(Q) < nval) {                                                                                    \
            const uint32_t vo = wd * (ls + (uint32_t)(Q));                                                             \
            x = 0xFFFFFFFFu;                                                                                           \
            if (vo + 8u <= room) {                                                                                     \
                /* bytes 4 .. wd - 2 hold neither a tab nor a newline (the bytes above read 0), byte wd - 1 is the   */ \
                /* tab (the newline behind the last sample), byte 3 the ':' that ends the GT                          */ \
                const uint32_t d1 = reinterpret_cast<const PlU32 *>(lbase + vo + 4u)->v;                               \
                const uint32_t low = d1 & lowm;                                                                        \
                const uint32_t t9 = low ^ 0x09090909u, tA = low ^ 0x0A0A0A0Au;                                         \
                uint32_t bad = (((t9 - 0x01010101u) & ~t9) | ((tA - 0x01010101u) & ~tA)) & 0x80808080u;                \
                bad |= ((d1 >> shb) & 0xFFu) ^ (ls + (uint32_t)(Q) == S_all - 1u ? 0x0Au : 0x09u);                     \
                const uint32_t d0 = reinterpret_cast<const PlU32 *>(lbase + vo)->v;                                    \
                bad |= (d0 >> 24) ^ (uint32_t)':';                                                                     \
                x = bad ? 0xFFFFFFFFu : ((d0 & 0x00FFFFFFu) | 0x09000000u);                                            \
            }                                                                                                          \
        }                                                                                                              \
        PL_CLASSIFY(x, Q);                                                                                             \
    } while (0)
            PL_STRIDED(0);
            PL_STRIDED(1);
            PL_STRIDED(2);
            PL_STRIDED(3);
#undef PL_STRIDED
        }
#undef PL_CLASSIFY
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(hard != 0u);
        if (bm != 0ull && lane == 0) {
            const uint32_t k = (uint32_t)(kbase + j);
            if (atomicExch(&redo_flag[k], 1u) == 0u) {
                unsigned long long slot = atomicAdd(&cnt->n_general, 1ull);
                redo_list[slot] = k;
            }
        }
    }
}

template <bool EDGE>
__device__ __forceinline__ void encode_planes_tile(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ k_soff,
                                                   const uint32_t *__restrict__ k_meta, uint64_t v_base, uint32_t n_kept,
                                                   const LayoutDev &lay, uint8_t *__restrict__ P, int8_t *__restrict__ G,
                                                   uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag,
                                                   DevCounters *cnt, uint32_t *img, uint32_t tile_v, uint32_t band)
{
    const uint64_t gv0 = (v_base / PT_V + tile_v) * (uint64_t)PT_V;   // first global column of the tile
    const long long k0 = (long long)gv0 - (long long)v_base;              // batch-local kept index of it
    if (k0 >= (long long)n_kept || (!lay.ring && gv0 >= lay.v_capacity)) return;
    const uint32_t s0 = band * TILE_S;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t S = lay.S;
    const uint32_t ls = s0 + 4u * lane;
    const uint32_t nval = ls >= S ? 0u : (S - ls >= 4u ? 4u : S - ls);
    const uint32_t last_q = (S - 1u >= ls && S - 1u < ls + 4u) ? S - 1u - ls : 4u;
    const long long kw = k0 + (long long)(w * PT_LW);

    constexpr int NG = 32 / PT_G;   // load groups per 32 lines
    PlStep nxt = pl_load_step(k_soff, k_meta, kw, n_kept, lane);
#pragma unroll 1
    for (int gp = 0; gp < PT_LW / 32; ++gp) {   // 32 lines -> one dword per (kind, plane, sample)
        const long long kb = kw + (long long)(gp * 32);
        const PlStep stp = nxt;
        if (gp + 1 < PT_LW / 32) nxt = pl_load_step(k_soff, k_meta, kb + 32, n_kept, lane);
        uint32_t o[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, e[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        // PT_SCHED(): the instruction scheduler may not move anything across — it would otherwise hoist the loads of all
        // four groups of a step to its top (128 registers of text in flight per lane, and spills)
#define PT_SCHED() __builtin_amdgcn_sched_barrier(0)
#if PT_DB
        PlGroup A, B;
        pl_load_group<EDGE>(A, text, n, stp, 0, ls, nval, last_q);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            pl_load_group<EDGE>(B, text, n, stp, (g + 1) * PT_G, ls, nval, last_q);
            PT_SCHED();
            pl_pack_group<EDGE>(A, (g * PT_G) & 15, o[(g * PT_G) >> 4], e[(g * PT_G) >> 4], kb + g * PT_G, text, n, stp, g * PT_G, ls, nval,
                                last_q, redo_list, redo_flag, cnt, lane, S);
            PT_SCHED();
            if (g + 2 < NG) pl_load_group<EDGE>(A, text, n, stp, (g + 2) * PT_G, ls, nval, last_q);
            PT_SCHED();
            pl_pack_group<EDGE>(B, ((g + 1) * PT_G) & 15, o[((g + 1) * PT_G) >> 4], e[((g + 1) * PT_G) >> 4], kb + (g + 1) * PT_G, text, n,
                                stp, (g + 1) * PT_G, ls, nval, last_q, redo_list, redo_flag, cnt, lane, S);
            PT_SCHED();
        }
#else
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            PlGroup A;
            pl_load_group<EDGE>(A, text, n, stp, g * PT_G, ls, nval, last_q);
            PT_SCHED();
            pl_pack_group<EDGE>(A, (g * PT_G) & 15, o[(g * PT_G) >> 4], e[(g * PT_G) >> 4], kb + g * PT_G, text, n, stp, g * PT_G, ls, nval,
                                last_q, redo_list, redo_flag, cnt, lane, S);
            PT_SCHED();
        }
#endif
        // image row (kind * 2 + plane) * 256 + sample, 8 dwords; dword c of a row sits at c ^ ((lane >> 1) & 7)
        const uint32_t c = (w * (uint32_t)(PT_LW / 32) + (uint32_t)gp) ^ ((lane >> 1) & 7u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t r = (4u * lane + (uint32_t)q) * PT_ROWDW + c;
            img[r] = __builtin_amdgcn_perm(o[1][q], o[0][q], 0x05040100u);                   // ONE, haplotype 0
            img[256u * PT_ROWDW + r] = __builtin_amdgcn_perm(o[1][q], o[0][q], 0x07060302u);   // ONE, haplotype 1
            img[512u * PT_ROWDW + r] = __builtin_amdgcn_perm(e[1][q], e[0][q], 0x05040100u);   // EXC, haplotype 0
            img[768u * PT_ROWDW + r] = __builtin_amdgcn_perm(e[1][q], e[0][q], 0x07060302u);   // EXC, haplotype 1
        }
    }
    __syncthreads();
    // image -> HBM: per kind-plane the 256 rows are one contiguous 8 KiB run of the tile (tile-major planes, common.h)
    uint64_t vcol = gv0 / lay.Vc;
    const uint64_t vin = gv0 - vcol * lay.Vc;
    if (lay.ring) vcol %= lay.ring;
    const PlanesGeom pg = planes_geom(lay, 0u);
    const uint32_t tile = (uint32_t)(vin / PL_TILE);
    const uint32_t nkeep = k0 < 0 ? (uint32_t)(-k0) : 0u;     // leading columns that belong to the previous call
#pragma unroll 4
    for (int it = 0; it < 2048 / (64 * PT_NW); ++it) {
        const uint32_t pi = (uint32_t)it * (64u * PT_NW) + threadIdx.x;   // 16-byte piece: kp * 512 + row * 2 + half
        const uint32_t kp = pi >> 9, row = (pi >> 1) & 255u, half = pi & 1u;
        const uint32_t s = s0 + row;
        if (s >= S) continue;
        const uint32_t sw = (row >> 3) & 7u;   // the writer's (lane >> 1) & 7: row = 4 lane + q
        const uint4 t = reinterpret_cast<const uint4 *>(img)[(kp * 256u + row) * 2u + (half ^ (sw >> 2))];
        uint32_t d0 = t.x, d1 = t.y, d2 = t.z, d3 = t.w;
        if (sw & 1u) {
            uint32_t u = d0; d0 = d1; d1 = u;
            u = d2; d2 = d3; d3 = u;
        }
        if (sw & 2u) {
            uint32_t u = d0; d0 = d2; d2 = u;
            u = d1; d1 = d3; d3 = u;
        }
        uint8_t *dst = P + planes_piece(pg, vcol, tile, kp, s) + half * 16u;
        if (nkeep) {   // (workgroup-uniform) the tile straddles the append position
            const uint32_t b0 = half * 128u;
            if (b0 + 128u <= nkeep) continue;
            if (b0 < nkeep) {
                const uint4 old = *reinterpret_cast<const uint4 *>(dst);
                const uint32_t nb = nkeep - b0;   // 1..127 low bits stay
                const uint32_t ov[4] = {old.x, old.y, old.z, old.w};
                uint32_t d[4] = {d0, d1, d2, d3};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t km = nb >= 32u * (i + 1) ? 0xFFFFFFFFu : (nb <= 32u * i ? 0u : ((1u << (nb - 32u * i)) - 1u));
                    d[i] = (ov[i] & km) | (d[i] & ~km);
                }
                d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
            }
        }
        __builtin_nontemporal_store(u32x4_al{d0, d1, d2, d3}, reinterpret_cast<u32x4_al *>(dst));
    }
}

// The band of sample columns that holds the last sample (and lanes past it) reads with care (EDGE); every other band reads
// 16 bytes flat.  One launch, a workgroup-uniform branch: since the medium path is a rolled loop both bodies fit the same
// 125 registers, and the last band's tiles fill the tail of the grid instead of being an under-filled launch of their own
// (round 3 first split them: 1.2 of 7.8 ms per step).
__global__ __launch_bounds__(64 * PT_NW, PT_WGS) void k_encode_planes(const uint8_t *__restrict__ text, uint64_t n,
                                                          const uint32_t *__restrict__ k_soff, const uint32_t *__restrict__ k_meta,
                                                          const uint64_t *__restrict__ d_cursor, LayoutDev lay,
                                                          uint8_t *__restrict__ P, int8_t *__restrict__ G,
                                                          uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag,
                                                          DevCounters *cnt, uint32_t tiles_v, uint32_t tiles_s, uint32_t map)
{
    HHGT_WAVE_PRIO();
    __shared__ __attribute__((aligned(16))) uint32_t img[1024 * PT_ROWDW];   // 32 KiB
    const uint64_t v_base = *d_cursor;
    const uint32_t n_kept = (uint32_t)cnt->n_kept;
    // which tile: (variant tile, sample band) from the linear workgroup id
    uint32_t tile_v, band;
    if (map == 0u) {          // variant tile fastest (round 3a's 2-D grid)
        tile_v = blockIdx.x % tiles_v;
        band = blockIdx.x / tiles_v;
    } else if (map == 1u) {   // sample band fastest: workgroups that run together read neighbouring KiB of the same lines
        band = blockIdx.x % tiles_s;
        tile_v = blockIdx.x / tiles_s;
    } else {                  // ... and the bands of one variant tile stay on one XCD (workgroup id mod 8), so the cache lines
        const uint32_t x = blockIdx.x & 7u, q = blockIdx.x >> 3;   // two neighbouring bands share are fetched by one L2
        band = q % tiles_s;
        tile_v = (q / tiles_s) * 8u + x;
    }
    if (tile_v >= tiles_v) return;
    if (band + 1u == tiles_s)
        encode_planes_tile<true>(text, n, k_soff, k_meta, v_base, n_kept, lay, P, G, redo_list, redo_flag, cnt, img, tile_v, band);
    else
        encode_planes_tile<false>(text, n, k_soff, k_meta, v_base, n_kept, lay, P, G, redo_list, redo_flag, cnt, img, tile_v, band);
}

// -------------------------------------------------------------------------------------------------
// htslib vcf_parse_format GT rule (see oracle/vcf_oracle.c parse_gt for the restatement it mirrors)
template <typename RD>
__device__ __forceinline__ uint32_t parse_gt_bytes(RD rd, uint32_t p, uint32_t lim, uint32_t *n_alleles)
{
    int l = 0;
    int vals[2] = {-9, -9};
    for (;;) {
        uint32_t c = p < lim ? rd(p) : 0u;
        if (c == '.') {
            ++p;
            if (l < 2) vals[l] = -9;
            ++l;
        } else if (c - '0' < 10u) {
            long long v = 0;
            while (p < lim) {
                uint32_t d = rd(p) - '0';
                if (d >= 10u) break;
                v = v * 10 + d;
                if (v > 0x7fffffffLL) v &= 0x7fffffffLL;
                ++p;
            }
            if (l < 2) vals[l] = (int)v;
            ++l;
        } else
            break;
        c = p < lim ? rd(p) : 0u;
        if (c != '|' && c != '/') break;
        ++p;
    }
    if (l == 0) {
        vals[0] = -9;
        l = 1;
    }
    *n_alleles = (uint32_t)l;
    return ((uint32_t)vals[0] & 0xFFu) | (((uint32_t)vals[1] & 0xFFu) << 8);
}

#define GEN_HALO 64u  // bytes staged past each 1 KiB piece: a GT sub-field that starts in the piece ends inside it

// 16 bytes of the text at `at` if they lie inside it, else what there is (zero-filled): the last line of a text may end
// within 16 bytes of the buffer's end
__device__ __forceinline__ uint4 gen_load16(const uint8_t *__restrict__ text, uint64_t n, uint64_t at)
{
    if (at + 16ull <= n) {
        const u32x4_unaligned t = *reinterpret_cast<const u32x4_unaligned *>(text + at);
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < 16u; ++k)
        if (at + k < n) w[k >> 2] |= (uint32_t)text[at + k] << (8u * (k & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// inclusive wave scan of small counts on DPP row shifts (no LDS round trips)
__device__ __forceinline__ uint32_t gen_scan_incl(uint32_t x, uint32_t lane)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31) + t0;
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + t1;
    const uint32_t row = lane >> 4;
    return x + (row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t1 : t2)));
}

// PLANES: the bit-plane form (k_encode_planes wrote zeros for every call of the lines this kernel owns; the bits are set
// by atomic or, the bytes of calls beyond 0 / 1 / missing go to their place in G).
// One wave per line, 1 KiB pieces.  Lane l holds bytes [16 l, 16 l + 16) of the piece and finds its tabs (one bit per
// byte); a wave scan numbers them, and every lane writes where the columns behind its tabs start into a list in LDS.
// Then the COLUMNS are dealt to the lanes, 64 consecutive samples per round: four bytes at the column's start from the
// staged piece, the branch-free {0, 1, .} x {|, /} classification of the tile kernel's second level (anything else:
// the byte walk of the GT rule, lane by lane), and the round's ONE / EXC bits of both haplotypes are four ballots —
// the nonzero calls go to the planes as global atomic ors (other lines of the same 32-variant group set their bits in the
// same dwords).
// Round 3, after config 4 spent 3.8 of 16 ms here (88 k `GT:DP` lines at 115 GB/s): the first version dealt the BYTES
// to the lanes and let every lane loop over the columns that start in its 16 (two or three half-empty iterations per
// piece); this one runs 3.6 ms, of which 2.8 without any global atomic (~500 instructions per KiB and wave).  Collecting a
// line's calls in an LDS image (four ballots per round, merged by three lanes) and sending them to the planes at the end of
// the line was built and measured: 4.3 ms — a burst of sparse device-scope atomics hides worse than a trickle (in git
// history: round 3).
template <bool PLANES>
__global__ __launch_bounds__(256) void k_encode_general(const uint8_t *__restrict__ text, uint64_t n,
                                                        const uint32_t *__restrict__ k_soff,
                                                        const uint32_t *__restrict__ k_lend,
                                                        const uint32_t *__restrict__ k_meta,
                                                        const uint32_t *__restrict__ redo_list,
                                                        const uint64_t *__restrict__ d_cursor,
                                                        LayoutDev lay, int8_t *__restrict__ G, uint8_t *__restrict__ P,
                                                        DevCounters *cnt)
{
    const uint64_t v_base = *d_cursor;
    const PlanesGeom pgeom = planes_geom(lay, 0u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t n_redo = (uint32_t)cnt->n_general;
    const uint32_t S = lay.S;
    __shared__ __attribute__((aligned(16))) uint8_t sbuf[4][1024 + GEN_HALO];   // the piece (plus a halo), per wave
    __shared__ uint16_t slist[4][1024];                                          // where its columns start
    // (the staged piece is always reached as sbuf[wq][...]: through a pointer variable the reader lambda below lost the LDS
    // address space and every character was a flat_load — 915 global-load instructions per line, 520 us per line)
    const uint32_t wq = threadIdx.x >> 6;
#define buf (sbuf[wq])
    uint32_t haploid = 0, malformed = 0, n_other = 0;
    for (uint32_t idx = wave; idx < n_redo; idx += n_waves) {
        const uint32_t k = redo_list[idx];
        const uint32_t soff = k_soff[k], lend = k_lend[k], gtidx = (k_meta[k] >> 8) & 0xFFFu;
        const uint64_t v = v_base + k;
        if (!lay.ring && v >= lay.v_capacity) continue;
        uint64_t vcol = v / lay.Vc;
        const uint64_t vin = v - vcol * lay.Vc;
        if (lay.ring) vcol %= lay.ring;
        // bit-plane form: where this line's bit lives in every sample's piece of the tile (wave-uniform)
        const uint64_t kstride = (uint64_t)pgeom.S_pad * 8ull;   // dwords between the kind-planes of a tile
        uint8_t *pline = PLANES ? P + planes_piece(pgeom, vcol, (uint32_t)(vin / PL_TILE), 0u, 0u) + (((uint32_t)(vin % PL_TILE) >> 5) << 2) : nullptr;
        const uint32_t pmask = 1u << ((uint32_t)(vin % PL_TILE) & 31u);
        const uint32_t rs = soff - 1u;  // the 9th tab: every sample field is preceded by a tab in [rs, lend)
        uint32_t tabs_before = 0, nl_inside = 0;
        // software pipeline: the piece in `cur` / `hal` was loaded one iteration ago
        uint4 cur = gen_load16(text, n, (uint64_t)rs + 16u * lane);
        uint4 hal = lane < GEN_HALO / 16u ? gen_load16(text, n, (uint64_t)rs + 1024u + 16u * lane) : make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t base = rs; base < lend; base += 1024u) {
            const uint32_t b0 = base + 16u * lane;
            const bool more = base + 1024u < lend;   // (wave-uniform)
            uint4 nx = make_uint4(0u, 0u, 0u, 0u), nh = nx;
            if (more) {
                nx = gen_load16(text, n, (uint64_t)b0 + 1024u);
                if (lane < GEN_HALO / 16u) nh = gen_load16(text, n, (uint64_t)base + 2048u + 16u * lane);
            }
            *reinterpret_cast<uint4 *>(buf + 16u * lane) = cur;
            if (lane < GEN_HALO / 16u) *reinterpret_cast<uint4 *>(buf + 1024u + 16u * lane) = hal;
            // exact per-byte "== tab" mask (SWAR), 4 bits per dword; bytes at or behind the line's end do not count
            uint32_t m = 0;
            {
                const uint32_t wv[4] = {cur.x, cur.y, cur.z, cur.w};
                const uint32_t inl = b0 >= lend ? 0u : (lend - b0 >= 16u ? 0xFFFFu : ((1u << (lend - b0)) - 1u));
                uint32_t nlm = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t t = wv[q] ^ 0x09090909u;
                    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
                    m |= ((((z >> 7) * 0x01020408u) >> 24) & 0xFu) << (4 * q);
                    const uint32_t tn = wv[q] ^ 0x0A0A0A0Au;
                    const uint32_t zn = ~(((tn & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | tn | 0x7F7F7F7Fu);
                    nlm |= ((((zn >> 7) * 0x01020408u) >> 24) & 0xFu) << (4 * q);
                }
                m &= inl;
                nl_inside |= nlm & inl;   // a newline in front of the line's own
            }
            const uint32_t avail = lend - base < 1024u + GEN_HALO ? lend - base : 1024u + GEN_HALO;
            // the staged piece through a pointer that CARRIES the LDS address space: as a generic pointer (or as one arm of a
            // select against the global text) every character the reader below fetches became a flat_load
            typedef __attribute__((address_space(3))) const uint8_t lds_u8;
            lds_u8 *lbuf = (lds_u8 *)(uintptr_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)sbuf[wq];
            auto rd = [&](uint32_t p) -> uint32_t {
                const uint32_t o = p - base;
                if (o < avail) return (uint32_t)lbuf[o];
                return (uint32_t)text[p];  // beyond the halo: rare long sub-fields
            };
            // the columns that start in this piece, in order: slist[i] = offset of the byte behind the i-th tab
            const uint32_t c = __popc(m);
            const uint32_t inc = gen_scan_incl(c, lane);
            const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            {
                uint32_t at = inc - c, mm = m;
                while (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
                    const bool on = mm != 0u;
                    const uint32_t j = (uint32_t)__builtin_ctz(mm | 0x10000u);
                    slist[wq][on ? at : 1023u] = (uint16_t)(16u * lane + j + 1u);   // (slot 1023 is never a column's: a piece with 1024 tabs has its last one at offset 1023, written by an `on` lane)
                    at += on ? 1u : 0u;
                    mm &= mm - 1u;
                }
            }
            // the lanes hand the staged piece and the column list to each other through LDS: the LDS executes a wave's
            // accesses in order, and these two lines keep the COMPILER from moving a lane's read of another lane's slot in
            // front of the stores above (no instruction is emitted; lz4bits.hip's BP_FENCE is the same pair)
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            for (uint32_t r0 = 0; r0 < nf; r0 += 64u) {   // (wave-uniform trip count)
                const uint32_t i = r0 + lane;
                const uint32_t s = tabs_before + i;
                const bool act = i < nf && s < S;   // surplus columns are not decoded (the oracle ignores them too)
                const uint32_t o0 = (uint32_t)slist[wq][i < 1024u ? i : 1023u];
                uint32_t p = base + (act ? o0 : 1u);
                bool missing = false;
                if (gtidx != 0u && act) {
                    for (uint32_t gsk = 0; gsk < gtidx; ++gsk) {  // skip to the GT sub-field
                        while (p < lend && rd(p) != ':' && rd(p) != '\t') ++p;
                        if (p < lend && rd(p) == ':') ++p;
                        else {
                            missing = true;
                            break;
                        }
                    }
                }
                uint32_t na = 2u, hv = 0xF7F7u;
                bool done = false;
                {
                    // the common column: GT comes first and reads "a|b" / "a/b" with a, b in {0, 1, .}, closed by ':', a tab or
                    // the line end.  Four bytes from the staged piece (two aligned dwords, one funnel shift), the same
                    // branch-free classification as the tile kernel's second level.
                    const bool inb = p - base + 8u <= avail;
                    const uint32_t o = inb ? p - base : 0u;
                    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(sbuf[wq]);
                    uint32_t x = __builtin_amdgcn_alignbyte(w32[(o >> 2) + 1u], w32[o >> 2], o & 3u);
                    if (p + 3u >= lend) x = (x & 0x00FFFFFFu) | 0x09000000u;   // the line (and the text) ends behind the call
                    const uint32_t y = x ^ 0x09307C30u;
                    const uint32_t yt = y >> 24;                                    // terminator: '\t', '\n' or ':'
                    const uint32_t t4 = y >> 4;
                    const uint32_t e2 = t4 & 0x00010001u, o2 = (y | t4) & 0x00010001u;
                    const uint32_t want = __umul24(e2, 30u) | (y & 0x00010001u & ~e2);
                    const uint32_t z = y & 0x0000FF00u, zz = z ^ 0x00005300u;
                    const uint32_t off = ((y & 0x00FF00FFu) ^ want) | (z < zz ? z : zz);
                    done = inb & !missing & (off == 0u) & ((yt == 0u) | (yt == 0x03u) | (yt == 0x33u)) & (p + 3u <= lend);
                    const uint32_t h0 = (e2 & 1u) ? 0xF7u : (o2 & 1u), h1 = (e2 >> 16) ? 0xF7u : (o2 >> 16);
                    hv = done ? (h0 | (h1 << 8)) : 0xF7F7u;
                }
                if (__builtin_amdgcn_ballot_w64(act && !done && !missing) != 0ull) {   // (wave-uniform: some lane needs the byte walk)
                    if (act && !done && !missing) {
                        na = 1u;
                        // the sub-field ends at ':' or at the column's tab; both stop the GT rule
                        hv = parse_gt_bytes(rd, p, lend, &na);
                    }
                }
                if (act && missing) na = 1u;   // a column without the GT sub-field counts as one missing allele (parse_gt_bytes of nothing)
                if (act && na == 1u) ++haploid;
                auto g_off = [&]() -> uint64_t {   // the call's place in the int8 matrix (64-bit multiplies: only where it is needed)
                    const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (s >> lay.sc_log2);
                    const uint32_t sin = s - scol * lay.Sc;
                    return (((vcol * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc + vin) * 2ull;
                };
                if (!PLANES) {
                    if (act) *reinterpret_cast<uint16_t *>(G + g_off()) = (uint16_t)hv;
                } else {
                    const uint32_t h0 = hv & 0xFFu, h1 = (hv >> 8) & 0xFFu;
                    // kind-planes of a tile: ONE of haplotype 0, ONE of haplotype 1, EXC of haplotype 0, EXC of haplotype 1
                    const bool bk[4] = {act && (h0 == 1u || h0 == 0xF7u), act && (h1 == 1u || h1 == 0xF7u), act && h0 > 1u, act && h1 > 1u};
#ifndef GEN_NO_SCATTER   // (development: timing of the column rounds alone; invalid planes)
                    {
                        // (pline: the line's tile and bit are the same for every sample — wave-uniform, computed once per line)
                        uint32_t *pw = reinterpret_cast<uint32_t *>(pline + (uint64_t)s * 32ull);
                        if (bk[0]) atomicOr(pw, pmask);
                        if (bk[1]) atomicOr(pw + kstride, pmask);
                        if (bk[2]) atomicOr(pw + 2ull * kstride, pmask);
                        if (bk[3]) atomicOr(pw + 3ull * kstride, pmask);
                    }
#endif
                    if (bk[2] && h0 != 0xF7u) {
                        ++n_other;
                        if (G) G[g_off()] = (int8_t)h0;
                    }
                    if (bk[3] && h1 != 0xF7u) {
                        ++n_other;
                        if (G) G[g_off() + 1ull] = (int8_t)h1;
                    }
                }
            }
            tabs_before += nf;
            cur = nx;
            hal = nh;
            __builtin_amdgcn_wave_barrier();   // ... and the next piece's stores behind this piece's reads
            asm volatile("" ::: "memory");
        }
        // fewer sample columns than the header declares; or a line end inside the line: a line shorter than any record
        // with S samples can be (its newline lay in the part the hopping index does not look at, index.hip)
        const bool broken = __builtin_amdgcn_ballot_w64(nl_inside != 0u) != 0ull;
        if ((tabs_before < S || broken) && lane == 0) ++malformed;
    }
    // one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        haploid += __shfl_down(haploid, d, 64);
        n_other += __shfl_down(n_other, d, 64);
    }
    if (lane == 0) {
        if (haploid) atomicAdd(&cnt->n_haploid, (unsigned long long)haploid);
        if (malformed) atomicAdd(&cnt->n_malformed, (unsigned long long)malformed);
        if (n_other) atomicAdd(&cnt->n_other, (unsigned long long)n_other);
    }
}
#undef buf

// -------------------------------------------------------------------------------------------------
// zero columns [c0, c1) (variant index inside the chunk column) of rows [r0, r1) (sample index inside
// the padded sample axis) of chunk column vcol
__global__ __launch_bounds__(256) void k_zero_rect(LayoutDev lay, uint64_t vcol0, uint32_t r0, uint32_t r1,
                                                   uint64_t c0, uint64_t c1, int8_t *__restrict__ G)
{
    const uint64_t vcol = vcol0 + blockIdx.z;   // one launch covers a range of chunk columns
    const uint32_t r = r0 + blockIdx.y;
    if (r >= r1) return;
    const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (r >> lay.sc_log2);
    const uint32_t sin = r - scol * lay.Sc;
    int8_t *row = G + (((vcol * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc) * 2ull;
    const uint64_t b0 = c0 * 2ull, b1 = c1 * 2ull;
    // 16-byte segments
    const uint64_t seg0 = b0 / 16ull, seg1 = (b1 + 15ull) / 16ull;
    for (uint64_t sgm = seg0 + blockIdx.x * blockDim.x + threadIdx.x; sgm < seg1;
         sgm += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = sgm * 16ull, hi = lo + 16ull;
        if (lo >= b0 && hi <= b1)
            *reinterpret_cast<uint4 *>(row + lo) = make_uint4(0, 0, 0, 0);
        else
            for (uint64_t b = lo < b0 ? b0 : lo; b < (hi > b1 ? b1 : hi); ++b) row[b] = 0;
    }
}

// cursor form of the tail padding: the kept count lives in device memory (asynchronous chains)
__global__ __launch_bounds__(256) void k_zero_tail_cursor(LayoutDev lay, const uint64_t *__restrict__ d_cursor,
                                                          int8_t *__restrict__ G)
{
    const uint64_t v_end = *d_cursor;
    uint64_t vcol = v_end / lay.Vc;
    const uint64_t c0 = v_end - vcol * lay.Vc;
    if (c0 == 0) return;                       // the cursor sits on a column boundary: nothing is open
    if (lay.ring) vcol %= lay.ring;
    const uint32_t r = blockIdx.y;             // padded sample row
    const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (r >> lay.sc_log2);
    const uint32_t sin = r - scol * lay.Sc;
    int8_t *row = G + (((vcol * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc) * 2ull;
    const uint64_t b0 = r < lay.S ? c0 * 2ull : 0ull, b1 = lay.Vc * 2ull;   // sample padding rows: the whole row
    const uint64_t seg0 = b0 / 16ull, seg1 = (b1 + 15ull) / 16ull;
    for (uint64_t sgm = seg0 + blockIdx.x * blockDim.x + threadIdx.x; sgm < seg1; sgm += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = sgm * 16ull, hi = lo + 16ull;
        if (lo >= b0 && hi <= b1)
            *reinterpret_cast<uint4 *>(row + lo) = make_uint4(0, 0, 0, 0);
        else
            for (uint64_t b = lo < b0 ? b0 : lo; b < (hi > b1 ? b1 : hi); ++b) row[b] = 0;
    }
}

int launch_pad_tail_cursor(LayoutDev lay, const uint64_t *d_cursor, int8_t *d_G, hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    if (S_pad == 0) return HHGT_OK;
    uint64_t segs = (lay.Vc * 2 + 15) / 16 + 1;
    uint32_t gx = (uint32_t)((segs + 255) / 256);
    if (gx > 8) gx = 8;
    hipLaunchKernelGGL(k_zero_tail_cursor, dim3(gx, S_pad), dim3(256), 0, st, lay, d_cursor, d_G);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// -------------------------------------------------------------------------------------------------
int launch_encode_tiles(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_meta,
                        uint32_t n_lines_bound, const uint64_t *d_cursor, LayoutDev lay, int8_t *d_G,
                        uint32_t *redo_list, uint32_t *redo_flag, DevCounters *d_cnt, hipStream_t st)
{
    if (n_lines_bound == 0 || lay.S == 0) return HHGT_OK;
    static const int tv = getenv("HHGT_TILE_V") ? atoi(getenv("HHGT_TILE_V")) : 64;
    // development: extra (unused) dynamic LDS per workgroup = fewer workgroups per CU (co-residency experiments)
    static const int enc_lds_pad = getenv("HHGT_ENC_LDS_PAD") ? atoi(getenv("HHGT_ENC_LDS_PAD")) : 0;
    uint32_t tiles_s = (lay.S + TILE_S - 1) / TILE_S;
    if (tv == 128) {
        // the append position is only known on the device: one tile more than the lines need covers any phase
        uint64_t tiles_v = (127 + (uint64_t)n_lines_bound + 127) / 128;
        hipLaunchKernelGGL(k_encode_tiles<128>, dim3((uint32_t)tiles_v, tiles_s), dim3(256), 0, st, d_text, n, k_soff,
                           k_meta, d_cursor, lay, d_G, redo_list, redo_flag, d_cnt);
    } else {
        uint64_t tiles_v = (63 + (uint64_t)n_lines_bound + 63) / 64;
        hipLaunchKernelGGL(k_encode_tiles<64>, dim3((uint32_t)tiles_v, tiles_s), dim3(256), enc_lds_pad, st, d_text, n, k_soff,
                           k_meta, d_cursor, lay, d_G, redo_list, redo_flag, d_cnt);
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_encode_general(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_lend,
                          const uint32_t *k_meta, const uint32_t *redo_list, const uint64_t *d_cursor, LayoutDev lay,
                          int8_t *d_G, uint8_t *d_P, DevCounters *d_cnt, int n_cu, hipStream_t st)
{
    if (lay.S == 0) return HHGT_OK;
    if (d_P)
        hipLaunchKernelGGL(k_encode_general<true>, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, d_text, n, k_soff, k_lend,
                           k_meta, redo_list, d_cursor, lay, d_G, d_P, d_cnt);
    else
        hipLaunchKernelGGL(k_encode_general<false>, dim3((uint32_t)n_cu * 8u), dim3(256), 0, st, d_text, n, k_soff, k_lend,
                           k_meta, redo_list, d_cursor, lay, d_G, d_P, d_cnt);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_encode_planes(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_meta,
                         uint32_t n_lines_bound, const uint64_t *d_cursor, LayoutDev lay, uint8_t *d_P, int8_t *d_G,
                         uint32_t *redo_list, uint32_t *redo_flag, DevCounters *d_cnt, hipStream_t st)
{
    if (n_lines_bound == 0 || lay.S == 0) return HHGT_OK;
    const uint32_t tiles_s = (lay.S + TILE_S - 1) / TILE_S;
    // the append position is only known on the device: one tile more than the lines need covers any phase
    const uint64_t tiles_v = ((uint64_t)(PT_V - 1) + (uint64_t)n_lines_bound + (PT_V - 1)) / PT_V;
    // measured (tools/dev/enc_map.sh, encode stage of the bench): 7.6 ms variant tile fastest, 7.4 ms band fastest / XCD-aware
    static const uint32_t map = getenv("HHGT_ENC_MAP") ? (uint32_t)atoi(getenv("HHGT_ENC_MAP")) : 2u;
    const uint64_t n_wg = map == 2u ? (tiles_v + 7ull) / 8ull * 8ull * tiles_s : tiles_v * tiles_s;
    hipLaunchKernelGGL(k_encode_planes, dim3((uint32_t)n_wg), dim3(64 * PT_NW), 0, st, d_text, n, k_soff, k_meta, d_cursor, lay,
                       d_P, d_G, redo_list, redo_flag, d_cnt, (uint32_t)tiles_v, tiles_s, map);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// -------------------------------------------------------------------------------------------------
// planes: zero the bits of variants [c0, Vc) of padded sample rows [r0, r1) of chunk columns vcol0 + blockIdx.z
// (tile-major planes: per tile and kind-plane the rows' 32-byte pieces are contiguous).  grid.x walks the tiles of the
// column, grid.y groups of 32 rows; a thread owns one dword of one row's piece, for the four kind-planes.
__device__ __forceinline__ void zero_planes_rows(const LayoutDev &lay, uint64_t vcol, uint32_t r0, uint32_t r1, uint64_t c0, uint32_t S,
                                                 uint8_t *__restrict__ P)
{
    const PlanesGeom pg = planes_geom(lay, 0u);
    const uint32_t r = r0 + blockIdx.y * 32u + (threadIdx.x >> 3), dw = threadIdx.x & 7u;
    if (r >= r1) return;
    const uint64_t from = r < S ? c0 : 0ull;   // sample padding rows: the whole row
    for (uint32_t t = blockIdx.x; t < pg.tpc; t += gridDim.x) {
        const uint64_t lo = (uint64_t)t * PL_TILE + 32ull * dw;   // first variant of this thread's dword
        if (lo + 32ull <= from) continue;
        const uint32_t keep = lo >= from ? 0u : ((1u << (uint32_t)(from - lo)) - 1u);
#pragma unroll
        for (uint32_t kp = 0; kp < 4u; ++kp) {
            uint32_t *p = reinterpret_cast<uint32_t *>(P + planes_piece(pg, vcol, t, kp, r)) + dw;
            *p = keep ? (*p & keep) : 0u;
        }
    }
}

__global__ __launch_bounds__(256) void k_zero_planes_rect(LayoutDev lay, uint64_t vcol0, uint32_t r0, uint32_t r1, uint64_t c0,
                                                          uint8_t *__restrict__ P)
{
    zero_planes_rows(lay, vcol0 + blockIdx.z, r0, r1, c0, 0xFFFFFFFFu, P);
}

__global__ __launch_bounds__(256) void k_zero_planes_cursor(LayoutDev lay, const uint64_t *__restrict__ d_cursor, uint8_t *__restrict__ P)
{
    const uint64_t v_end = *d_cursor;
    uint64_t vcol = v_end / lay.Vc;
    const uint64_t c0 = v_end - vcol * lay.Vc;
    if (c0 == 0) return;                       // the cursor sits on a column boundary: nothing is open
    if (lay.ring) vcol %= lay.ring;
    zero_planes_rows(lay, vcol, 0u, lay.n_sc * lay.Sc, c0, lay.S, P);
}

int launch_pad_tail_planes_cursor(LayoutDev lay, const uint64_t *d_cursor, uint8_t *d_P, hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    if (S_pad == 0) return HHGT_OK;
    const uint32_t tpc = (uint32_t)(lay.Vc / PL_TILE);
    hipLaunchKernelGGL(k_zero_planes_cursor, dim3(tpc < 32 ? tpc : 32, (S_pad + 31) / 32), dim3(256), 0, st, lay, d_cursor, d_P);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_pad_tail_planes(LayoutDev lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end, uint8_t *d_P, hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    const uint32_t tpc = (uint32_t)(lay.Vc / PL_TILE);
    const uint32_t gx = tpc < 32 ? tpc : 32;
    if (v_end % lay.Vc) {   // (a) variant padding of the last touched chunk column
        const uint64_t vcol = v_end / lay.Vc;
        if (vcol < vcol_end && vcol >= vcol_begin && S_pad)
            hipLaunchKernelGGL(k_zero_planes_rect, dim3(gx, (S_pad + 31) / 32), dim3(256), 0, st, lay, vcol, 0u, S_pad, v_end - vcol * lay.Vc, d_P);
    }
    if (S_pad > lay.S) {    // (b) sample padding rows of every touched chunk column
        for (uint64_t v0 = vcol_begin; v0 < vcol_end; v0 += 65535) {
            const uint32_t nz = (uint32_t)(vcol_end - v0 < 65535 ? vcol_end - v0 : 65535);
            hipLaunchKernelGGL(k_zero_planes_rect, dim3(gx, (S_pad - lay.S + 31) / 32, nz), dim3(256), 0, st, lay, v0, lay.S, S_pad, 0ull, d_P);
        }
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// -------------------------------------------------------------------------------------------------
// planes -> int8 bytes (the inverse of the packing; for consumers that want the matrix after all and for the parity
// tests).  One workgroup per Blosc block (one sample row x 4096 variants = 16 tiles), thread t owns variants
// [16 t, 16 t + 16): 2 bytes of each kind-plane piece of tile t / 16.
__global__ __launch_bounds__(256) void k_planes_expand(LayoutDev lay, PlanesGeom pg, const uint8_t *__restrict__ P, const uint8_t *G,
                                                       uint8_t *out, uint64_t id0)   // out may be G
{
    uint64_t col;
    uint32_t row, bi;
    planes_block(pg, id0 + blockIdx.x, &col, &row, &bi);
    const uint32_t t = threadIdx.x;
    const uint32_t tile = bi * 16u + (t >> 4), sub = t & 15u;   // 16-bit word `sub` of the tile's 32-byte piece
    const uint64_t kstride = (uint64_t)pg.S_pad * 16ull;        // 16-bit words between kind-planes
    const uint16_t *pl = reinterpret_cast<const uint16_t *>(P + planes_piece(pg, col, tile, 0u, row)) + sub;
    const uint32_t o0 = pl[0], o1 = pl[kstride], e0 = pl[2ull * kstride], e1 = pl[3ull * kstride];
    // byte offset of the block inside the int8 matrix (chunk-tiled layout: column, sample chunk, row in chunk)
    const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (row >> lay.sc_log2);
    const uint32_t sin = row - scol * lay.Sc;
    const uint64_t base = ((((col * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc) + (uint64_t)bi * 4096ull + 16ull * t) * 2ull;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 2 * i + k;
            uint32_t h0 = (o0 >> j) & 1u, h1 = (o1 >> j) & 1u;
            if ((e0 >> j) & 1u) h0 = h0 ? 0xF7u : (G ? (uint32_t)G[base + 2u * j] : 0u);
            if ((e1 >> j) & 1u) h1 = h1 ? 0xF7u : (G ? (uint32_t)G[base + 2u * j + 1u] : 0u);
            x |= (h0 | (h1 << 8)) << (16 * k);
        }
        w[i] = x;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + base);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

int launch_planes_expand(LayoutDev lay, const uint8_t *d_P, const uint8_t *d_G, uint32_t col0, uint32_t n_cols, uint8_t *d_out, hipStream_t st)
{
    const PlanesGeom pg = planes_geom(lay, col0);
    const uint64_t n_blocks = (uint64_t)n_cols * pg.S_pad * pg.bpr;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += 0x40000000ull) {
        const uint64_t nb = n_blocks - b0 < 0x40000000ull ? n_blocks - b0 : 0x40000000ull;
        hipLaunchKernelGGL(k_planes_expand, dim3((uint32_t)nb), dim3(256), 0, st, lay, pg, d_P, d_G, d_out, b0);
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
int launch_pad_tail(LayoutDev lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end, int8_t *d_G,
                    hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    // (a) variant padding of the last touched chunk column
    if (v_end % lay.Vc) {
        uint64_t vcol = v_end / lay.Vc;
        uint64_t c0 = v_end - vcol * lay.Vc, c1 = lay.Vc;
        if (vcol < vcol_end && vcol >= vcol_begin && S_pad) {
            uint64_t segs = ((c1 - c0) * 2 + 15) / 16 + 1;
            uint32_t gx = (uint32_t)((segs + 255) / 256);
            if (gx > 1024) gx = 1024;
            hipLaunchKernelGGL(k_zero_rect, dim3(gx, S_pad), dim3(256), 0, st, lay, vcol, 0u, S_pad, c0, c1, d_G);
        }
    }
    // (b) sample padding rows of every touched chunk column
    if (S_pad > lay.S) {
        uint64_t segs = (lay.Vc * 2 + 15) / 16 + 1;
        uint32_t gx = (uint32_t)((segs + 255) / 256);
        if (gx > 64) gx = 64;
        for (uint64_t v0 = vcol_begin; v0 < vcol_end; v0 += 65535) {
            uint32_t nz = (uint32_t)(vcol_end - v0 < 65535 ? vcol_end - v0 : 65535);
            hipLaunchKernelGGL(k_zero_rect, dim3(gx, S_pad - lay.S, nz), dim3(256), 0, st, lay, v0, lay.S, S_pad, 0ull,
                               lay.Vc, d_G);
        }
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
