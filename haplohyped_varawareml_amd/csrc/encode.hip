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
// A 256-thread workgroup owns 512 variants x 256 samples: 512 variants so that every (sample, plane) leaves as 64
// contiguous bytes; wave w reads lines [128 w, 128 w + 128) of the tile, 8 lines (1 KiB each, 16 B per lane = 4
// samples) in flight while the previous 8 are packed.  Packing is SWAR on the field dword "a|b\t": (x & 0x00010001)
// holds the two allele bits, shifted by the line number they accumulate into a register whose low half is haplotype 0
// and whose high half is haplotype 1 over 16 lines; one xor/or chain per line tells whether all four fields of the lane
// were "[01]|[01]\t".  Only a group of 16 lines in which some lane saw anything else (a missing call, '/', a third
// allele) is parsed again field by field.  Per 32 lines a lane holds one dword per (sample, plane, kind): 16
// conflict-free ds_write_b32 into a 64 KiB image [kind][plane][sample][64 B]; after one barrier the image leaves in
// 16-byte pieces, four lanes per 64-byte row.  The tile that straddles the append position merges with the bits the
// previous call wrote.
// HBM roofline: algorithmic bytes per variant = 4 S read + S/2 written.
#define PT_V 512
#define PT_LW 128
#define PT_G 8
#define PT_C 0x09307C30u   // "0|0\t"

struct PlGroup {
    uint4 raw[PT_G];
    uint32_t lvalid;   // bit j: line j of the group is a kept fixed-width line (wave-uniform)
};

template <bool EDGE>
__device__ __forceinline__ void pl_load_group(PlGroup &gr, const uint8_t *__restrict__ text, uint64_t n,
                                              const uint32_t *__restrict__ k_soff, const uint32_t *__restrict__ k_meta,
                                              long long kbase, uint32_t n_kept, uint32_t ls, uint32_t nval, uint32_t last_q)
{
    uint32_t lv = 0;
#pragma unroll
    for (int j = 0; j < PT_G; ++j) {
        const long long k = kbase + j;
        bool valid = k >= 0 && k < (long long)n_kept;
        uint32_t meta = 0, soff = 0;
        if (valid) {
            meta = k_meta[k];
            soff = k_soff[k];
        }
        valid = valid && (meta & LF_FAST);
        uint4 v = make_uint4(FILL_FIELD, FILL_FIELD, FILL_FIELD, FILL_FIELD);
        if (valid) {
            lv |= 1u << j;
            const uint64_t off = (uint64_t)soff + 4ull * ls;
            if (!EDGE) {   // every lane owns four samples in front of the line's last one: the 16 bytes lie inside the line
                u32x4_unaligned t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(text + off));
                v = make_uint4(t.x, t.y, t.z, t.w);
            } else if (nval == 4u && off + 16ull <= n) {
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
            if (EDGE) {   // the last sample of a line is terminated by the line end (LF_FAST pinned where it ends)
                if (last_q == 0u) v.x = (v.x & 0x00FFFFFFu) | 0x09000000u;
                if (last_q == 1u) v.y = (v.y & 0x00FFFFFFu) | 0x09000000u;
                if (last_q == 2u) v.z = (v.z & 0x00FFFFFFu) | 0x09000000u;
                if (last_q == 3u) v.w = (v.w & 0x00FFFFFFu) | 0x09000000u;
            }
        }
        gr.raw[j] = v;
    }
    gr.lvalid = lv;
}

// 16 lines -> per sample q: one[q] / exc[q], bit j (haplotype 0) and bit 16 + j (haplotype 1) for line j.
// Three levels.  (1) every field of every lane is "[01]|[01]\t": 14 vector instructions per line.  (2) some lane saw
// something else: the group again with every field classified over the alphabet {0, 1, .} x {|, /} — still branch-free
// SWAR on the field dword, x ^ "0|0\t": an allele byte must be 0x00, 0x01 or 0x1E ('.'), bit 4 of it is the EXC bit and
// bit 0 | bit 4 the ONE bit; the separator byte 0x00 or 0x53 ('/'), the terminator 0x00.  (3) a field outside that
// alphabet (a third allele, a multi-digit index, anything malformed) contributes zero bits and sends its line to the
// variable-width kernel, which sets the bits of the whole line (idempotent for the fields that were fine here).
__device__ __forceinline__ void pl_pack_group(const PlGroup &gr, const int sh, uint32_t (&one)[4], uint32_t (&exc)[4], long long kbase,
                                              uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag, DevCounters *cnt,
                                              uint32_t lane)
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
    if (__builtin_amdgcn_ballot_w64((bad & 0xFFFEFFFEu) != 0u) == 0ull) {
        one[0] |= a0, one[1] |= a1, one[2] |= a2, one[3] |= a3;
        return;
    }
#pragma unroll
    for (int j = 0; j < PT_G; ++j) {
        const uint32_t xs[4] = {gr.raw[j].x, gr.raw[j].y, gr.raw[j].z, gr.raw[j].w};
        uint32_t hard = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t y = xs[q] ^ PT_C;
            const uint32_t t4 = y >> 4;
            const uint32_t e2 = t4 & 0x00010001u;                                   // '.' in either allele
            const uint32_t o2 = (y | t4) & 0x00010001u;                             // '1' or '.'
            const uint32_t want = __umul24(e2, 30u) | (y & 0x00010001u & ~e2);      // what the allele bytes must then be
            const uint32_t z = y & 0xFF00FF00u, zz = z ^ 0x00005300u;               // separator / terminator bytes
            const uint32_t off = ((y & 0x00FF00FFu) ^ want) | (z < zz ? z : zz);
            const uint32_t keep = off ? 0u : 0xFFFFFFFFu;
            one[q] |= (o2 & keep) << (sh + j);
            exc[q] |= (e2 & keep) << (sh + j);
            hard |= off;
        }
        if ((gr.lvalid >> j) & 1u) {
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
}

template <bool EDGE>
__device__ __forceinline__ void encode_planes_tile(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ k_soff,
                                                   const uint32_t *__restrict__ k_meta, uint64_t v_base, uint32_t n_kept,
                                                   const LayoutDev &lay, uint8_t *__restrict__ P, int8_t *__restrict__ G,
                                                   uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag,
                                                   DevCounters *cnt, uint32_t *img)
{
    const uint64_t gv0 = (v_base / PT_V + blockIdx.x) * (uint64_t)PT_V;   // first global column of the tile
    const long long k0 = (long long)gv0 - (long long)v_base;              // batch-local kept index of it
    if (k0 >= (long long)n_kept || (!lay.ring && gv0 >= lay.v_capacity)) return;
    const uint32_t s0 = blockIdx.y * TILE_S;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t S = lay.S;
    const uint32_t ls = s0 + 4u * lane;
    const uint32_t nval = ls >= S ? 0u : (S - ls >= 4u ? 4u : S - ls);
    const uint32_t last_q = (S - 1u >= ls && S - 1u < ls + 4u) ? S - 1u - ls : 4u;
    const long long kw = k0 + (long long)(w * PT_LW);

    PlGroup A, B;
    pl_load_group<EDGE>(A, text, n, k_soff, k_meta, kw, n_kept, ls, nval, last_q);
#pragma unroll 1
    for (int gp = 0; gp < PT_LW / 32; ++gp) {   // 32 lines = four groups of 8, the next group's loads in flight while one is packed
        const long long kb = kw + (long long)(gp * 32);
        uint32_t o0[4] = {0, 0, 0, 0}, e0[4] = {0, 0, 0, 0}, o1[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0};
        pl_load_group<EDGE>(B, text, n, k_soff, k_meta, kb + 8, n_kept, ls, nval, last_q);
        pl_pack_group(A, 0, o0, e0, kb, redo_list, redo_flag, cnt, lane);
        pl_load_group<EDGE>(A, text, n, k_soff, k_meta, kb + 16, n_kept, ls, nval, last_q);
        pl_pack_group(B, 8, o0, e0, kb + 8, redo_list, redo_flag, cnt, lane);
        pl_load_group<EDGE>(B, text, n, k_soff, k_meta, kb + 24, n_kept, ls, nval, last_q);
        pl_pack_group(A, 0, o1, e1, kb + 16, redo_list, redo_flag, cnt, lane);
        if (gp + 1 < PT_LW / 32) pl_load_group<EDGE>(A, text, n, k_soff, k_meta, kb + 32, n_kept, ls, nval, last_q);
        pl_pack_group(B, 8, o1, e1, kb + 24, redo_list, redo_flag, cnt, lane);
        // 32 lines of this lane's four samples: one dword per (kind, plane, sample).  Image row (kind * 2 + plane) * 256
        // + q * 64 + lane, 16 dwords; dword c of a row sits at c ^ (lane >> 2): the 64 lanes of a store hit 64 banks.
        const uint32_t c = (w * 4u + (uint32_t)gp) ^ (lane >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t r = ((uint32_t)q * 64u + lane) * 16u + c;
            img[r] = __builtin_amdgcn_perm(o1[q], o0[q], 0x05040100u);             // ONE, haplotype 0
            img[256u * 16u + r] = __builtin_amdgcn_perm(o1[q], o0[q], 0x07060302u);   // ONE, haplotype 1
            img[512u * 16u + r] = __builtin_amdgcn_perm(e1[q], e0[q], 0x05040100u);   // EXC, haplotype 0
            img[768u * 16u + r] = __builtin_amdgcn_perm(e1[q], e0[q], 0x07060302u);   // EXC, haplotype 1
        }
    }
    __syncthreads();
    // image -> HBM: 1024 rows x 64 B, four lanes per row
    uint64_t vcol = gv0 / lay.Vc;
    const uint64_t vin = gv0 - vcol * lay.Vc;
    if (lay.ring) vcol %= lay.ring;
    const uint64_t bpr = lay.Vc >> 12;                       // Blosc blocks per sample row of a chunk
    const uint32_t piece = (uint32_t)((vin & 4095ull) >> 9);  // which 64 bytes of the block's planes
    const uint32_t nkeep = k0 < 0 ? (uint32_t)(-k0) : 0u;     // leading columns that belong to the previous call
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const uint32_t pi = (uint32_t)it * 256u + threadIdx.x;
        const uint32_t R = pi >> 2, sl = pi & 3u;
        const uint32_t kp = R >> 8, q = (R >> 6) & 3u, l = R & 63u;
        const uint32_t s = s0 + 4u * l + q;
        if (s >= S) continue;
        const uint32_t sw = l >> 2;
        const uint4 t = reinterpret_cast<const uint4 *>(img)[R * 4u + (sl ^ (sw >> 2))];
        uint32_t d0 = t.x, d1 = t.y, d2 = t.z, d3 = t.w;
        if (sw & 1u) {
            uint32_t u = d0; d0 = d1; d1 = u;
            u = d2; d2 = d3; d3 = u;
        }
        if (sw & 2u) {
            uint32_t u = d0; d0 = d2; d2 = u;
            u = d1; d1 = d3; d3 = u;
        }
        const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (s >> lay.sc_log2);
        const uint32_t sin = s - scol * lay.Sc;
        const uint64_t blk = ((vcol * lay.n_sc + scol) * lay.Sc + sin) * bpr + (vin >> 12);
        uint8_t *dst = P + blk * 2048ull + kp * 512u + piece * 64u + sl * 16u;
        if (nkeep) {   // (workgroup-uniform) the tile straddles the append position
            const uint32_t b0 = sl * 128u;
            if (b0 + 128u <= nkeep) continue;
            if (b0 < nkeep) {
                const uint4 old = *reinterpret_cast<const uint4 *>(dst);
                const uint32_t nb = nkeep - b0;   // 1..127 low bits stay
                const uint32_t o[4] = {old.x, old.y, old.z, old.w};
                uint32_t d[4] = {d0, d1, d2, d3};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t km = nb >= 32u * (i + 1) ? 0xFFFFFFFFu : (nb <= 32u * i ? 0u : ((1u << (nb - 32u * i)) - 1u));
                    d[i] = (o[i] & km) | (d[i] & ~km);
                }
                d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
            }
        }
        __builtin_nontemporal_store(u32x4_al{d0, d1, d2, d3}, reinterpret_cast<u32x4_al *>(dst));
    }
}

__global__ __launch_bounds__(256, 2) void k_encode_planes(const uint8_t *__restrict__ text, uint64_t n,
                                                          const uint32_t *__restrict__ k_soff, const uint32_t *__restrict__ k_meta,
                                                          const uint64_t *__restrict__ d_cursor, LayoutDev lay,
                                                          uint8_t *__restrict__ P, int8_t *__restrict__ G,
                                                          uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag,
                                                          DevCounters *cnt)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[1024 * 16];   // 64 KiB
    const uint64_t v_base = *d_cursor;
    const uint32_t n_kept = (uint32_t)cnt->n_kept;
    // the tile row that holds the last sample (and lanes past it) reads with care; every other tile reads 16 bytes flat
    if (blockIdx.y + 1u == gridDim.y)
        encode_planes_tile<true>(text, n, k_soff, k_meta, v_base, n_kept, lay, P, G, redo_list, redo_flag, cnt, img);
    else
        encode_planes_tile<false>(text, n, k_soff, k_meta, v_base, n_kept, lay, P, G, redo_list, redo_flag, cnt, img);
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

// PLANES: the bit-plane form (k_encode_planes wrote zeros for every call of the lines this kernel owns; the bits are set
// by atomic or, the bytes of calls beyond 0 / 1 / missing go to their place in G)
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
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t n_redo = (uint32_t)cnt->n_general;
    const uint32_t S = lay.S;
    // each wave stages the 1 KiB piece it is ranking (plus a halo) in LDS: the per-field byte walk then reads LDS
    // instead of issuing one global load per character
    __shared__ __attribute__((aligned(16))) uint8_t sbuf[4][1024 + GEN_HALO];
    uint8_t *buf = sbuf[threadIdx.x >> 6];
    uint32_t haploid = 0, malformed = 0;
    for (uint32_t idx = wave; idx < n_redo; idx += n_waves) {
        const uint32_t k = redo_list[idx];
        const uint32_t soff = k_soff[k], lend = k_lend[k], gtidx = k_meta[k] >> 8;
        const uint64_t v = v_base + k;
        if (!lay.ring && v >= lay.v_capacity) continue;
        uint64_t vcol = v / lay.Vc;
        const uint64_t vin = v - vcol * lay.Vc;
        if (lay.ring) vcol %= lay.ring;
        const uint32_t rs = soff - 1u;  // the 9th tab: every sample field is preceded by a tab in [rs, lend)
        uint32_t tabs_before = 0, nl_inside = 0;
        for (uint32_t base = rs; base < lend; base += 1024u) {
            const uint32_t b0 = base + 16u * lane;
            uint32_t m = 0;
            if (b0 + 16u <= lend) {
                // one unaligned 16-byte load + exact per-byte "== tab" mask (SWAR), 4 bits per dword
                const u32x4_unaligned t4 = *reinterpret_cast<const u32x4_unaligned *>(text + b0);
                const uint32_t wv[4] = {t4.x, t4.y, t4.z, t4.w};
                *reinterpret_cast<uint4 *>(buf + 16u * lane) = make_uint4(t4.x, t4.y, t4.z, t4.w);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t t = wv[q] ^ 0x09090909u;
                    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
                    m |= ((((z >> 7) * 0x01020408u) >> 24) & 0xFu) << (4 * q);
                    const uint32_t tn = wv[q] ^ 0x0A0A0A0Au;
                    nl_inside |= (tn - 0x01010101u) & ~tn & 0x80808080u;   // a newline in front of the line's own
                }
            } else if (b0 < lend) {
                for (uint32_t j = 0; j < 16u; ++j) {
                    uint32_t p = b0 + j;
                    if (p < lend) {
                        const uint8_t ch = text[p];
                        buf[16u * lane + j] = ch;
                        if (ch == '\t') m |= 1u << j;
                        if (ch == '\n') nl_inside = 1u;
                    }
                }
            }
            if (lane < GEN_HALO / 16u) {
                const uint32_t hb = base + 1024u + 16u * lane;
                if (hb + 16u <= lend) {
                    const u32x4_unaligned h4 = *reinterpret_cast<const u32x4_unaligned *>(text + hb);
                    *reinterpret_cast<uint4 *>(buf + 1024u + 16u * lane) = make_uint4(h4.x, h4.y, h4.z, h4.w);
                } else {
                    for (uint32_t j = 0; j < 16u; ++j)
                        if (hb + j < lend) buf[1024u + 16u * lane + j] = text[hb + j];
                }
            }
            const uint32_t avail = lend - base < 1024u + GEN_HALO ? lend - base : 1024u + GEN_HALO;
            auto rd = [&](uint32_t p) -> uint32_t {
                const uint32_t o = p - base;
                return o < avail ? (uint32_t)buf[o] : (uint32_t)text[p];  // beyond the halo: rare long sub-fields
            };
            uint32_t c = __popc(m), inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(inc, d, 64);
                if (lane >= (uint32_t)d) inc += t;
            }
            uint32_t f = tabs_before + inc - c;
            while (m) {
                int j = __ffs(m) - 1;
                m &= m - 1;
                const uint32_t s = f++;
                if (s >= S) continue;  // surplus columns are not decoded (the oracle ignores them too)
                uint32_t p = b0 + (uint32_t)j + 1u;
                bool missing = false;
                for (uint32_t gsk = 0; gsk < gtidx; ++gsk) {  // skip to the GT sub-field
                    while (p < lend && rd(p) != ':' && rd(p) != '\t') ++p;
                    if (p < lend && rd(p) == ':') ++p;
                    else {
                        missing = true;
                        break;
                    }
                }
                uint32_t na = 1, hv = 0xF7F7u;
                if (!missing) {
                    // the sub-field ends at ':' or at the column's tab; both stop the GT rule
                    hv = parse_gt_bytes(rd, p, lend, &na);
                }
                if (na == 1u) ++haploid;
                const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (s >> lay.sc_log2);
                const uint32_t sin = s - scol * lay.Sc;
                const uint64_t goff = (((vcol * lay.n_sc + scol) * lay.Sc + sin) * lay.Vc + vin) * 2ull;
                if (!PLANES) {
                    *reinterpret_cast<uint16_t *>(G + goff) = (uint16_t)hv;
                } else {
                    const uint32_t bit = (uint32_t)(goff & 8191ull) >> 1;
                    uint32_t *pw = reinterpret_cast<uint32_t *>(P + (goff >> 13) * 2048ull) + (bit >> 5);
                    const uint32_t mk = 1u << (bit & 31u);
                    const uint32_t h0 = hv & 0xFFu, h1 = (hv >> 8) & 0xFFu;
                    if (h0 == 1u || h0 == 0xF7u) atomicOr(pw, mk);
                    if (h1 == 1u || h1 == 0xF7u) atomicOr(pw + 128, mk);
                    if (h0 > 1u) {
                        atomicOr(pw + 256, mk);
                        if (h0 != 0xF7u) {
                            atomicAdd(&cnt->n_other, 1ull);
                            if (G) G[goff] = (int8_t)h0;
                        }
                    }
                    if (h1 > 1u) {
                        atomicOr(pw + 384, mk);
                        if (h1 != 0xF7u) {
                            atomicAdd(&cnt->n_other, 1ull);
                            if (G) G[goff + 1ull] = (int8_t)h1;
                        }
                    }
                }
            }
            tabs_before += __shfl(inc, 63, 64);
        }
        // fewer sample columns than the header declares; or a line end inside the line: a line shorter than any record
        // with S samples can be (its newline lay in the part the hopping index does not look at, index.hip)
        const bool broken = __builtin_amdgcn_ballot_w64(nl_inside != 0u) != 0ull;
        if ((tabs_before < S || broken) && lane == 0) ++malformed;
    }
    // one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) haploid += __shfl_down(haploid, d, 64);
    if (lane == 0) {
        if (haploid) atomicAdd(&cnt->n_haploid, (unsigned long long)haploid);
        if (malformed) atomicAdd(&cnt->n_malformed, (unsigned long long)malformed);
    }
}

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
    hipLaunchKernelGGL(k_encode_planes, dim3((uint32_t)tiles_v, tiles_s), dim3(256), 0, st, d_text, n, k_soff, k_meta, d_cursor,
                       lay, d_P, d_G, redo_list, redo_flag, d_cnt);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// -------------------------------------------------------------------------------------------------
// planes: zero the bits of variants [c0, Vc) of padded sample rows [r0, r1) of chunk columns vcol0 + blockIdx.z
// (rows >= S: the whole row).  grid.x = Blosc blocks per sample row, 128 threads = the 128 dwords of a plane.
__device__ __forceinline__ void zero_planes_row(const LayoutDev &lay, uint64_t vcol, uint32_t r, uint64_t c0, uint8_t *__restrict__ P)
{
    const uint32_t scol = lay.sc_log2 >= 31u ? 0u : (r >> lay.sc_log2);
    const uint32_t sin = r - scol * lay.Sc;
    const uint64_t bpr = lay.Vc >> 12;
    for (uint64_t b = blockIdx.x; b < bpr; b += gridDim.x) {
        const uint64_t lo = b * 4096ull + 32ull * threadIdx.x;   // first variant of this thread's dword
        if (lo + 32ull <= c0) continue;
        const uint32_t keep = lo >= c0 ? 0u : ((1u << (uint32_t)(c0 - lo)) - 1u);
        uint32_t *p = reinterpret_cast<uint32_t *>(P + (((vcol * lay.n_sc + scol) * lay.Sc + sin) * bpr + b) * 2048ull) + threadIdx.x;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) p[128 * kp] = keep ? (p[128 * kp] & keep) : 0u;
    }
}

__global__ __launch_bounds__(128) void k_zero_planes_rect(LayoutDev lay, uint64_t vcol0, uint32_t r0, uint32_t r1, uint64_t c0,
                                                          uint8_t *__restrict__ P)
{
    const uint32_t r = r0 + blockIdx.y;
    if (r >= r1) return;
    zero_planes_row(lay, vcol0 + blockIdx.z, r, c0, P);
}

__global__ __launch_bounds__(128) void k_zero_planes_cursor(LayoutDev lay, const uint64_t *__restrict__ d_cursor, uint8_t *__restrict__ P)
{
    const uint64_t v_end = *d_cursor;
    uint64_t vcol = v_end / lay.Vc;
    const uint64_t c0 = v_end - vcol * lay.Vc;
    if (c0 == 0) return;                       // the cursor sits on a column boundary: nothing is open
    if (lay.ring) vcol %= lay.ring;
    const uint32_t r = blockIdx.y;             // padded sample row
    zero_planes_row(lay, vcol, r, r < lay.S ? c0 : 0ull, P);
}

int launch_pad_tail_planes_cursor(LayoutDev lay, const uint64_t *d_cursor, uint8_t *d_P, hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    if (S_pad == 0) return HHGT_OK;
    const uint64_t bpr = lay.Vc >> 12;
    hipLaunchKernelGGL(k_zero_planes_cursor, dim3((uint32_t)(bpr < 64 ? bpr : 64), S_pad), dim3(128), 0, st, lay, d_cursor, d_P);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_pad_tail_planes(LayoutDev lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end, uint8_t *d_P, hipStream_t st)
{
    const uint32_t S_pad = lay.n_sc * lay.Sc;
    const uint64_t bpr = lay.Vc >> 12;
    const uint32_t gx = (uint32_t)(bpr < 64 ? bpr : 64);
    if (v_end % lay.Vc) {   // (a) variant padding of the last touched chunk column
        const uint64_t vcol = v_end / lay.Vc;
        if (vcol < vcol_end && vcol >= vcol_begin && S_pad)
            hipLaunchKernelGGL(k_zero_planes_rect, dim3(gx, S_pad), dim3(128), 0, st, lay, vcol, 0u, S_pad, v_end - vcol * lay.Vc, d_P);
    }
    if (S_pad > lay.S) {    // (b) sample padding rows of every touched chunk column
        for (uint64_t v0 = vcol_begin; v0 < vcol_end; v0 += 65535) {
            const uint32_t nz = (uint32_t)(vcol_end - v0 < 65535 ? vcol_end - v0 : 65535);
            hipLaunchKernelGGL(k_zero_planes_rect, dim3(gx, S_pad - lay.S, nz), dim3(128), 0, st, lay, v0, lay.S, S_pad, 0ull, d_P);
        }
    }
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

// -------------------------------------------------------------------------------------------------
// planes -> int8 bytes of the block (the inverse of the packing; for consumers that want the matrix after all and for
// the parity tests).  One workgroup per Blosc block, thread t owns variants [16 t, 16 t + 16).
__global__ __launch_bounds__(256) void k_planes_expand(const uint8_t *__restrict__ P, const uint8_t *G, uint8_t *out)   // out may be G
{
    const uint64_t blk = blockIdx.x;
    const uint16_t *pl = reinterpret_cast<const uint16_t *>(P + blk * 2048ull);
    const uint32_t t = threadIdx.x;
    const uint32_t o0 = pl[t], o1 = pl[256 + t], e0 = pl[512 + t], e1 = pl[768 + t];
    const uint64_t base = blk * 8192ull + 32ull * t;
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

int launch_planes_expand(const uint8_t *d_P, const uint8_t *d_G, uint64_t n_blocks, uint8_t *d_out, hipStream_t st)
{
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += 0x40000000ull) {
        const uint64_t nb = n_blocks - b0 < 0x40000000ull ? n_blocks - b0 : 0x40000000ull;
        hipLaunchKernelGGL(k_planes_expand, dim3((uint32_t)nb), dim3(256), 0, st, d_P + b0 * 2048ull, d_G ? d_G + b0 * 8192ull : nullptr,
                           d_out + b0 * 8192ull);
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
