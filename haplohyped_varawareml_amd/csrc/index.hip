// index.hip — line framing and fixed-column parse on device-resident VCF text (gfx950).
//
// Replaces, for a whole block of text at once, what the reference does one record at a time:
//   tbx_itr_next -> raw text line                 /root/reference/cpp/vcfpp.h:1468
//   vcf_parse1: CHROM POS ID REF ALT ... FORMAT   /root/reference/cpp/vcfpp.h:1471  (htslib)
//   BcfRecord::isSNP filter                       /root/reference/cpp/vcfpp.h:990-1000
//   Start()/End()/REF()/ALT()                     /root/reference/cpp/vcfpp.h:1118-1133,1142-1151
//   region restriction (tabix contig[:beg-end])   /root/reference/cpp/vcfpp.h:1424-1451
//
// Stage 1 (k_index_newlines): ONE streaming read of the text, 16 B per lane, coalesced; each wave
//   owns a 16 KiB region and records the newline positions it finds in a private slot array.
//   HBM bound: algorithmic bytes = nbytes read + 4 B per line written.
// Stage 2 (k_compact_newlines): prefix over per-region counts -> dense nl[] array (O(lines)).
// Stage 3 (k_parse_fixed): one lane per line walks the 9 fixed columns (O(F) bytes per line,
//   F ~ 60-300 B, versus the ~10 KB sample region that only the encode stage reads).
// Stage 4 (k_compact_kept): kept records are compacted (exclusive scan of keep flags) and the
//   per-record table (start, stop, ref, alt) is written at v_base + k.
#include "common.h"

typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t nl_mask4(uint32_t x)
{
    // exact per-byte "== '\n'" -> bit k set for byte k
    uint32_t t = x ^ 0x0A0A0A0Au;
    uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);  // 0x80 where byte == 0
    return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
}

// grid: ceil(n_regions / 4) blocks of 256 threads; wave w of block b scans region 4b + w.
__global__ __launch_bounds__(256) void k_index_newlines(const uint8_t *__restrict__ text, uint64_t n,
                                                        uint32_t *__restrict__ slots,
                                                        uint32_t *__restrict__ counts, uint32_t n_regions,
                                                        DevCounters *cnt)
{
    HHGT_WAVE_PRIO();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= n_regions) return;
    // a text that does not end in '\n' gets a virtual newline at position n
    const bool virt = n > 0 && text[n - 1] != '\n';
    const uint64_t rbase = (uint64_t)r * INDEX_REGION;
    uint32_t *my = slots + (size_t)r * INDEX_CAP;
    uint32_t wcount = 0;
    bool overflow = false;
    uint32_t masks[INDEX_REGION / 1024];
    // issue all loads first (16 independent 16 B loads per lane), then resolve
#pragma unroll
    for (int it = 0; it < (int)(INDEX_REGION / 1024); ++it) {
        uint64_t g0 = rbase + (uint64_t)it * 1024u + lane * 16u;
        uint32_t m = 0;
        if (g0 + 16 <= n) {
            const u32x4_t v4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(text + g0));  // streamed once
            uint4 v = make_uint4(v4.x, v4.y, v4.z, v4.w);
            m = nl_mask4(v.x) | (nl_mask4(v.y) << 4) | (nl_mask4(v.z) << 8) | (nl_mask4(v.w) << 12);
        } else if (g0 <= n) {
            for (int j = 0; j < 16; ++j) {
                uint64_t g = g0 + j;
                if (g < n) {
                    if (text[g] == '\n') m |= 1u << j;
                } else if (g == n && virt)
                    m |= 1u << j;
            }
        }
        masks[it] = m;
    }
#pragma unroll
    for (int it = 0; it < (int)(INDEX_REGION / 1024); ++it) {
        uint32_t m = masks[it];
        if (__ballot(m != 0) == 0ull) continue;  // wave-uniform: long lines mostly skip
        uint32_t c = __popc(m);
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += t;
        }
        uint32_t idx = wcount + inc - c;
        uint32_t g0 = (uint32_t)(rbase + (uint64_t)it * 1024u + lane * 16u);
        while (m) {
            int j = __ffs(m) - 1;
            m &= m - 1;
            if (idx < INDEX_CAP) my[idx] = g0 + (uint32_t)j;
            else overflow = true;
            ++idx;
        }
        wcount += __shfl(inc, 63, 64);
    }
    if (__ballot(overflow) != 0ull && lane == 0) atomicAdd(&cnt->err_density, 1ull);
    if (lane == 0) counts[r] = wcount < INDEX_CAP ? wcount : INDEX_CAP;
}

// ---------------------------------------------------------------------------------------------
// k_index_hop<U, WALK>: the same index for text whose lines cannot be shorter than `skip` bytes — a record with S sample
// columns has at least 2 S + 17 bytes in front of its newline, so behind every line start that many bytes need not be
// looked at.  A wave owns HOP_K (<= 64, the launcher's choice) consecutive regions and walks them line by line, appending
// every newline to the slot list of the region it lies in (same slots / counts layout as k_index_newlines: everything
// downstream is unchanged).  The first newline of a wave's range is found by plain scanning, so ranges need no hand-over.
//
// WALK = false (round 2, HHGT_INDEX_MODE=1): from `pos` the wave loads U KiB (16 B per lane, all loads in flight
// together), takes the first newline, looks at the first byte of the next line ('#' header lines and empty lines are
// short: no skip behind them) and hops by the bound — for the 1000G shape (lines of 4 S + 58 bytes) it reads half of
// the text.
//
// WALK = true (round 4, the default): behind a newline the wave loads the HEAD of the next line — 1 KiB, 16 B per lane,
// starting 15 bytes in front of the line so that lane 0 holds the newline itself — ranks the tabs in it (one 16-bit mask
// per lane, a DPP prefix count, two ballots) and thereby knows where the sample columns start.  If FORMAT is exactly
// "GT", a record of S diploid calls "a|b" has its newline at soff + 4 S - 1: the next head is loaded THERE, and when its
// byte 15 is the newline the line has cost one load of 1 KiB instead of a search through U KiB (and nothing of the
// sample columns was read: the index pass reads ~1.1 KB per 10 KB line).  Anything else — another FORMAT, a byte that
// is not a newline, a '\r' in front of it, fewer than nine tabs in the head (an INFO column of more than ~900 bytes), a
// newline inside the head, the last KiB of the text — falls back to the search, from the tighter bound soff + 2 S - 1
// where the head gave one.  What the walk does NOT look at are sample columns, exactly like the hop: a line that is
// shorter than its head claims is malformed either way; when its newline falls into a hop it merges with the next line,
// and the newline is found where the merged line is read: by the encoders inside the sample columns of a KEPT record
// (every field is matched against "a|b\t", encode.hip), by k_parse_fixed in the skipped bytes of a record the filters
// DROP (nobody else reads those) — the same error, one record later.
__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t c4)
{
    // exact per-byte "== c" -> bit k set for byte k (c4 = the byte in all four positions)
    const uint32_t t = x ^ c4;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);  // 0x80 where byte == 0
    return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
}

__device__ __forceinline__ uint32_t eq_mask16(const u32x4_t &v, uint32_t c4)
{
    return eq_mask4(v.x, c4) | (eq_mask4(v.y, c4) << 4) | (eq_mask4(v.z, c4) << 8) | (eq_mask4(v.w, c4) << 12);
}

// byte `off` (wave-uniform, < 1024) of a head window held 16 bytes per lane
__device__ __forceinline__ uint32_t head_byte(const u32x4_t &v, uint32_t off)
{
    const uint32_t q = (off >> 2) & 3u;
    const uint32_t d = q == 0u ? v.x : (q == 1u ? v.y : (q == 2u ? v.z : v.w));
    return ((uint32_t)__builtin_amdgcn_readlane((int)d, (int)(off >> 4)) >> (8u * (off & 3u))) & 0xFFu;
}

// offset of the k-th (1-based) set bit of the wave's 1024-bit mask (16 bits per lane, lane order), given the exclusive
// prefix count `cum` of the lanes' popcounts; the caller knows that there are at least k
__device__ __forceinline__ uint32_t head_kth(uint32_t m16, uint32_t cum, uint32_t k)
{
    const unsigned long long b = __builtin_amdgcn_ballot_w64(cum < k && cum + (uint32_t)__popc(m16) >= k);
    const int l = __builtin_ctzll(b);
    uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)m16, l);
    const uint32_t r = k - (uint32_t)__builtin_amdgcn_readlane((int)cum, l);
    for (uint32_t i = 1; i < r; ++i) m &= m - 1u;
    return 16u * (uint32_t)l + (uint32_t)__builtin_ctz(m);
}

#ifndef WALK_BATCH
#define WALK_BATCH 1   // four lines per step (below); 0: one line per step
#endif
#ifndef WALK_PF
#define WALK_PF 0   // 1: two head windows in flight per wave (built and measured in round 4: index stage 2.78 against 2.40 ms — the
                    // walk is not a latency chain at the occupancy it runs at; kept for the record)
#endif
template <int U, bool WALK>
__global__ __launch_bounds__(256) void k_index_hop(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ slots,
                                                   uint32_t *__restrict__ counts, uint32_t n_regions, uint32_t skip, uint32_t S,
                                                   DevCounters *cnt, uint32_t HOP_K)
{
    HHGT_WAVE_PRIO();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    const uint32_t r_first = w * HOP_K;
    if (r_first >= n_regions) return;
    const uint32_t r_cnt = n_regions - r_first < HOP_K ? n_regions - r_first : HOP_K;
    const bool virt = n > 0 && text[n - 1] != '\n';   // a text that does not end in '\n' gets a virtual newline at n
    const uint64_t beg = (uint64_t)r_first * INDEX_REGION;
    uint64_t endw = beg + (uint64_t)r_cnt * INDEX_REGION;   // positions [beg, endw) are this wave's; position n counts
    if (endw > n + 1) endw = n + 1;
    const uint64_t last_term = virt ? n : n - 1;
    uint32_t cntv = 0;        // lane i: newlines found in region r_first + i
    bool overflow = false;
    uint64_t pos = beg;
    // WALK: the position the head of the last line says its newline is at (have_cand), and where to search if it is not
    uint64_t cand = 0, fb = 0;
    bool have_cand = false;
    // WALK: a line is one dependent load, so a wave is a latency chain — while the window at `cand` is on its way, the one
    // behind the NEXT line is requested too, where that line's newline will be if the two lines are of one length (their
    // fixed columns differ by a few bytes at most: the window starts 47 bytes in front of the guess and is taken when the
    // newline turns out to lie in its bytes 15 .. 79).  A wrong guess costs nothing but the load.
    uint64_t est_len = 0, pf_wb = 0;
    bool pf_valid = false;
    u32x4_t pf = u32x4_t{0u, 0u, 0u, 0u};
    // appends newline q (< endw) to the slot list of its region
    auto record = [&](uint64_t q) {
        const uint32_t rr = (uint32_t)(q / INDEX_REGION), local = rr - r_first;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)local);
        if (c < INDEX_CAP) {
            if (lane == 0) slots[(size_t)rr * INDEX_CAP + c] = (uint32_t)q;
        } else
            overflow = true;
        cntv += lane == local ? 1u : 0u;
    };
    // WALK, four lines per step (WALK_BATCH): the walk is bound by the ~300 instructions a wave issues per line, not by the
    // latency of its one load (two windows in flight changed nothing, WALK_PF) — so the same instructions serve FOUR lines:
    // 16 lanes x 16 bytes each look at a 256-byte window where the newline of line k + g should be if the lines are of one
    // length (40 bytes in front of the guess), find the newline, rank the tabs behind it (row-wise DPP scan), and the lane
    // that holds the ninth tab checks "\tGT\t" and says where the NEXT newline must be.  Then the chain is verified on the
    // scalar side: group g's newline must sit exactly where group g - 1 said (group 0: where the candidate is).  The first
    // link that does not hold — another width, another FORMAT, a '\r', a head of more than ~190 bytes, a '#' line — hands
    // over to the one-line path below, which decides with its 1 KiB head or a search; two steps in a row that get nowhere
    // switch the batches off for the rest of the wave's range.
    bool no_batch = false, bat_off = false;
    uint32_t bat_fail = 0;
    // A batch reads 256 bytes of a head where the one-line path reads 1 KiB — and a record that is too short for its S
    // samples but ends inside that KiB is recognised there (its newline is IN the head).  So a candidate that came out of
    // a batch and turns out not to be a newline sends the wave back to the head it came from, once, with the wide window.
    uint64_t len_other = 0, srch_soff = 0;   // bytes of the sample columns of the last record whose FORMAT was not "GT"; start of the one being searched
    uint64_t head_p = 0;       // the (verified) newline in front of the line whose head produced `cand`
    bool cand_batch = false;   // ... and that head was read by a batch
    bool rehead = false;
    while (have_cand || pos < endw) {
        if (WALK && WALK_BATCH && have_cand && !no_batch && !bat_off && est_len != 0ull && cand >= 40ull && cand + 3ull * est_len + 216ull <= n) {
            const uint32_t g = lane >> 4, li = lane & 15u;
            const uint64_t L0 = est_len;   // (the windows lie where THIS length puts them; est_len moves on with the chain)
            const uint8_t *wbase = text + (cand - 40ull);
            const uint64_t voff = (uint64_t)g * L0 + 16ull * li;
            const u32x4_t v = *reinterpret_cast<const u32x4_unaligned *>(wbase + voff);
            const uint32_t nlm = eq_mask16(v, 0x0A0A0A0Au), tbm = eq_mask16(v, 0x09090909u);
            const unsigned long long nlb = __builtin_amdgcn_ballot_w64(nlm != 0u);
            uint32_t og[4];
            bool vg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t mg = (uint32_t)(nlb >> (16 * k)) & 0xFFFFu;
                vg[k] = mg != 0u;
                const int fl = 16 * k + __builtin_ctz(mg | 0x10000u);
                og[k] = 16u * ((uint32_t)fl & 15u) + (uint32_t)__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)nlm, fl & 63) | 0x10000u);
            }
            const uint32_t o = g == 0u ? og[0] : (g == 1u ? og[1] : (g == 2u ? og[2] : og[3]));
            const uint32_t l0 = 16u * li;
            const uint32_t keep = l0 > o ? 0xFFFFu : (l0 + 15u <= o ? 0u : (0xFFFFu << (o - l0 + 1u)) & 0xFFFFu);
            const unsigned long long nl2 = __builtin_amdgcn_ballot_w64((nlm & keep) != 0u);   // a second newline: the line ends inside its head
            const uint32_t tk = tbm & keep, tc = (uint32_t)__popc(tk);
            uint32_t incl = tc;   // inclusive count inside the row of 16 lanes
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
            const uint32_t cum = incl - tc;
            const bool has9 = cum < 9u && incl >= 9u;
            // in the lane of the ninth tab: its offset, "\tGT" in front of it, the next newline
            uint32_t m9 = tk;
            const uint32_t r1 = 8u - cum;   // lower tabs of this lane to step over (has9: 0 .. 8)
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) m9 = i < r1 ? (m9 & (m9 - 1u)) : m9;
            const uint32_t j9 = (uint32_t)__builtin_ctz(m9 | 0x10000u) & 15u;   // byte of the lane
            const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x111, 0xf, 0xf, false);      // previous lane's last dword
            const uint32_t ptb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tbm, 0x111, 0xf, 0xf, false);    // ... and its tab mask
            const uint32_t tb20 = (tbm << 4) | (ptb >> 12);           // tab bits of the 20 bytes [previous lane's last four | own 16]
            const bool tab8 = ((tb20 >> (j9 + 1u)) & 1u) != 0u;       // byte j9 - 3 of the lane
            const uint32_t ix = j9 + 2u, di = ix >> 2;                // bytes j9 - 2, j9 - 1 in the 20-byte array
            const uint32_t w0 = di == 0u ? pw : (di == 1u ? v.x : (di == 2u ? v.y : (di == 3u ? v.z : v.w)));
            const uint32_t w1 = di == 0u ? v.x : (di == 1u ? v.y : (di == 2u ? v.z : v.w));
            const uint32_t gt2 = __builtin_amdgcn_alignbyte(w1, w0, ix & 3u) & 0xFFFFu;
            const bool ok9 = has9 && tab8 && gt2 == 0x5447u;           // 'G' | 'T' << 8
            const uint32_t t9 = l0 + j9;
            const unsigned long long b9 = __builtin_amdgcn_ballot_w64(ok9);
            // ---- the chain, on the scalar side
            uint64_t expected = cand, fbx = fb, exp_head = head_p;
            bool exp_batch = cand_batch;
            bool stop = false, handed = false;
            uint32_t advanced = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (stop || handed) continue;
                const uint64_t nlpos = cand - 40ull + (uint64_t)k * L0 + og[k];
                const bool cr = vg[k] && og[k] != 0u && head_byte(v, 256u * (uint32_t)k + og[k] - 1u) == '\r';
                if (!vg[k] || nlpos != expected || cr) {   // not where it should be (or a CRLF text): the one-line path decides
                    handed = true;
                    continue;
                }
                exp_head = expected;
                exp_batch = true;
                srch_soff = 0ull;
                if (expected >= endw) {
                    stop = true;
                    continue;
                }
                const uint32_t bits = (uint32_t)(b9 >> (16 * k)) & 0xFFFFu;
                if (bits == 0u || ((nl2 >> (16 * k)) & 0xFFFFull) != 0ull) {
                    // the newline is there, the head behind it is not of this shape: found again (and recorded) by the search
                    have_cand = false;
                    pos = expected;
                    handed = true;
                    expected = ~0ull;
                    continue;
                }
                record(expected);
                ++advanced;
                const int l9 = 16 * k + __builtin_ctz(bits);
                const uint32_t t9s = (uint32_t)__builtin_amdgcn_readlane((int)t9, l9);
                const uint64_t soff = cand - 40ull + (uint64_t)k * L0 + t9s + 1ull;
                const uint64_t nxt = soff + 4ull * S - 1ull;
                est_len = nxt - expected;
                expected = nxt;
                fbx = soff + 2ull * S - 17ull;
            }
            if (stop) break;
            if (expected != ~0ull) {   // the next candidate (unverified), for the next step or for the one-line path
                if (expected > last_term) {   // (a line that claims to reach past the text: the search decides)
                    have_cand = false;
                    pos = fbx;
                    if (pos > last_term) pos = last_term;
                } else {
                    cand = expected;
                    fb = fbx > last_term ? last_term : fbx;
                    have_cand = true;
                }
                head_p = exp_head;
                cand_batch = exp_batch;
            }
            no_batch = handed;
            bat_fail = advanced ? 0u : bat_fail + 1u;
            bat_off = bat_fail >= 2u;
            continue;
        }
        no_batch = false;
        bool found = false, nb_have = false, have_head = false;
        uint32_t nb_reg = 0, ho = 15u;   // ho: offset of the newline inside the head window
        uint64_t p = 0;
        u32x4_t hd = u32x4_t{0u, 0u, 0u, 0u};
        if (WALK && have_cand) {
            have_cand = false;
            uint32_t c15 = 0, c14 = 0;
            const bool use_pf = pf_valid && cand >= pf_wb + 15ull && cand <= pf_wb + 79ull;
            pf_valid = false;
            if (use_pf || cand + 1009ull <= n) {   // the window [cand - 15, cand + 1009) lies inside the text
                if (use_pf) {
                    hd = pf;
                    ho = (uint32_t)(cand - pf_wb);
                } else
                    hd = *reinterpret_cast<const u32x4_unaligned *>(text + (cand - 15ull) + 16ull * lane);
                const uint64_t pred = cand + est_len;   // the newline behind the line that starts behind `cand`
                if (WALK_PF && est_len != 0ull && pred + 977ull <= n && cand < endw) {
                    pf_wb = pred - 47ull;
                    pf = *reinterpret_cast<const u32x4_unaligned *>(text + pf_wb + 16ull * lane);
                    pf_valid = true;
                }
                c15 = head_byte(hd, ho);
                c14 = head_byte(hd, ho - 1u);
                have_head = true;
            } else if (cand < n) {
                c15 = text[cand];
                c14 = text[cand - 1];
            } else if (cand == n && virt) {
                c15 = 0x0Au;
                c14 = text[n - 1];
            }
            // ('\r' in front: bgzf_getline strips it and the sample columns are one byte shorter — not this shape)
            if (c15 == 0x0Au && c14 != '\r') {
                found = true;
                p = cand;
                srch_soff = 0ull;
            } else if (cand_batch) {
                // the head this candidate came from, once more and 1 KiB wide (no newline is recorded: head_p already is)
                cand_batch = false;
                rehead = true;
                found = true;
                have_head = false;
                p = head_p;
            } else {
                pos = fb;
                continue;
            }
        } else {
        const uint64_t a0 = pos & ~15ull;
        uint32_t any[U];
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t g0 = a0 + (uint64_t)u * 1024u + lane * 16u;
            uint32_t f = 0;
            v[u] = u32x4_t{0u, 0u, 0u, 0u};
            if (g0 + 16 <= n) {
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(text + g0));   // streamed once
                const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t t = d[q] ^ 0x0A0A0A0Au;
                    f |= (t - 0x01010101u) & ~t & 0x80808080u;   // non-zero iff one of the four bytes is '\n'
                }
            } else if (g0 <= n) {
                // the last bytes of the text: assembled byte by byte, position n holds the virtual newline
                uint32_t d[4] = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
                for (int j = 0; j < 16; ++j) {
                    const uint64_t g = g0 + j;
                    uint32_t ch = 0x20u;
                    if (g < n) ch = text[g];
                    else if (g == n && virt) ch = 0x0Au;
                    d[j >> 2] = (d[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | (ch << (8 * (j & 3)));
                    if (ch == 0x0Au) f = 1u;
                }
                v[u] = u32x4_t{d[0], d[1], d[2], d[3]};
            }
            // bytes in front of pos are not part of the search
            if (g0 + 16 <= pos) f = 0;
            any[u] = f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (found) continue;
            unsigned long long b = __builtin_amdgcn_ballot_w64(any[u] != 0u);
            while (b != 0ull && !found) {
                // the exact mask of the first flagged lane (its flag may stem from bytes in front of pos only)
                const int l = __builtin_ctzll(b);
                b &= b - 1ull;
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)v[u].x, l), x1 = (uint32_t)__builtin_amdgcn_readlane((int)v[u].y, l);
                const uint32_t x2 = (uint32_t)__builtin_amdgcn_readlane((int)v[u].z, l), x3 = (uint32_t)__builtin_amdgcn_readlane((int)v[u].w, l);
                uint32_t m = nl_mask4(x0) | (nl_mask4(x1) << 4) | (nl_mask4(x2) << 8) | (nl_mask4(x3) << 12);
                const uint64_t g0 = a0 + (uint64_t)u * 1024u + (uint64_t)l * 16u;
                if (g0 < pos) m &= ~((1u << (uint32_t)(pos - g0)) - 1u);
                if (m) {
                    found = true;
                    const uint32_t j = (uint32_t)__builtin_ctz(m);
                    p = g0 + j;
                    if (j < 15u && g0 + 16 <= n) {
                        const uint32_t k = j + 1u, dw = k >> 2;
                        const uint32_t xd = dw == 0u ? x0 : (dw == 1u ? x1 : (dw == 2u ? x2 : x3));
                        nb_reg = (xd >> (8u * (k & 3u))) & 0xFFu;
                        nb_have = true;
                    }
                }
            }
        }
        if (!found) {
            pos = a0 + (uint64_t)U * 1024u;
            continue;
        }
        if (WALK && srch_soff != 0ull) {   // the sample columns of a record with another FORMAT: as long as the next one's, maybe
            len_other = p - srch_soff;
            srch_soff = 0ull;
        }
        }
        if (!rehead) {
            if (p >= endw) break;
            record(p);
        } else
            no_batch = true;   // (the candidate the wide head gives is verified by the one-line path)
        rehead = false;
        bool hopped = false;
        if (WALK) {
            // the head of the line behind the newline (a search found p: one more load; the candidate's window is it already)
            if (!have_head && p >= 15ull && p + 1009ull <= n) {
                hd = *reinterpret_cast<const u32x4_unaligned *>(text + (p - 15ull) + 16ull * lane);
                have_head = true;
            }
            if (have_head) {
                const uint64_t wb = p - (uint64_t)ho;
                const uint32_t nb = head_byte(hd, ho + 1u);
                // the lane's bytes that lie behind the newline (offsets > ho): only those belong to the head
                const uint32_t l0 = 16u * lane;
                const uint32_t keep = l0 > ho ? 0xFFFFu : (l0 + 15u <= ho ? 0u : (0xFFFFu << (ho - l0 + 1u)) & 0xFFFFu);
                if (nb == '#' || nb == 0x0Au) {
                    pos = p + 1;   // header and empty lines are short: search right behind them
                    hopped = true;
                } else {
                    const uint32_t nlm = eq_mask16(hd, 0x0A0A0A0Au) & keep;
                    const unsigned long long nlb = __builtin_amdgcn_ballot_w64(nlm != 0u);
                    if (nlb != 0ull) {
                        // the line ends inside its head (far too short for S samples): exactly there
                        const int l = __builtin_ctzll(nlb);
                        pos = wb + 16ull * (uint64_t)l + (uint64_t)__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)nlm, l));
                        hopped = true;
                    } else {
                        const uint32_t tbm = eq_mask16(hd, 0x09090909u) & keep;
                        const uint32_t tc = (uint32_t)__popc(tbm);
                        const uint32_t incl = wave_scan_sum_dpp(tc, lane);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        if (total >= 9u) {
                            const uint32_t t8 = head_kth(tbm, incl - tc, 8u), t9 = head_kth(tbm, incl - tc, 9u);
                            const uint64_t soff = wb + t9 + 1ull;
                            // S fields of at least one byte and S - 1 tabs in front of the newline (16 bytes of margin as in `skip`)
                            const uint64_t lo = soff + 2ull * S - 17ull;
                            const bool gt = t9 - t8 == 3u && head_byte(hd, t8 + 1u) == 'G' && head_byte(hd, t8 + 2u) == 'T';
                            if (gt && soff + 4ull * S - 1ull <= last_term) {
                                cand = soff + 4ull * S - 1ull;
                                fb = lo;
                                have_cand = true;
                                est_len = cand - p;
                                head_p = p;
                                cand_batch = false;
                            } else if (!gt && len_other != 0ull && soff + len_other <= last_term) {
                                // another FORMAT ("GT:DP" ...): no width to compute — but such records often repeat the width
                                // of the last one of their kind (fixed-width sub-fields): the newline is tried there, the
                                // search from the bound stays the fallback, the encoders read every byte of a kept record
                                cand = soff + len_other;
                                fb = lo;
                                have_cand = true;
                                head_p = p;
                                cand_batch = false;
                                srch_soff = soff;
                            } else {
                                pos = lo;
                                srch_soff = gt ? 0ull : soff;
                            }
                            hopped = true;
                        }
                    }
                }
            }
        }
        if (!hopped) {
            uint32_t nb = 0x0Au;
            if (nb_have) nb = nb_reg;                 // the byte behind the newline came with the same 16 bytes (15 of 16 cases)
            else if (p + 1 < n) nb = text[p + 1];
            // ('\r': the empty line of a CRLF text — round 3 hopped behind it and merged it with the record that follows)
            pos = p + 1 + ((nb == '#' || nb == 0x0Au || nb == '\r') ? 0u : skip);
        }
        // the terminator of the text's last line is always looked at: a file cut off in mid-line ends in a line shorter
        // than the bound, which has to reach the parser (and be reported), not vanish in a hop
        if (pos > last_term) pos = p + 1 > last_term ? p + 1 : last_term;
        if (have_cand && fb > last_term) fb = p + 1 > last_term ? p + 1 : last_term;
    }
    if (overflow && lane == 0) atomicAdd(&cnt->err_density, 1ull);
    if (lane < r_cnt) counts[r_first + lane] = cntv < INDEX_CAP ? cntv : INDEX_CAP;
}

// one thread per region: copy its slot entries to their final place
__global__ __launch_bounds__(256) void k_compact_newlines(const uint32_t *__restrict__ slots,
                                                          const uint32_t *__restrict__ counts,
                                                          const uint32_t *__restrict__ prefix,
                                                          uint32_t n_regions, uint32_t *__restrict__ nl,
                                                          uint32_t max_lines)
{
    HHGT_WAVE_PRIO();
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    uint32_t c = counts[r], p = prefix[r];
    const uint32_t *my = slots + (size_t)r * INDEX_CAP;
    for (uint32_t k = 0; k < c && p + k < max_lines; ++k) nl[p + k] = my[k];   // lines past the caller's bound: err_lines
}

// ---------------------------------------------------------------------------------------------
// fixed columns: one lane per line
__global__ __launch_bounds__(256) void k_parse_fixed(const uint8_t *__restrict__ text, uint64_t n,
                                                     const uint32_t *__restrict__ nl, const uint32_t *__restrict__ d_nlines,
                                                     uint32_t max_lines, const RegionFilter rfv, uint32_t S,
                                                     uint32_t *__restrict__ l_soff, uint32_t *__restrict__ l_lend,
                                                     uint32_t *__restrict__ l_pos, uint32_t *__restrict__ l_refalt,
                                                     uint32_t *__restrict__ l_flags, uint32_t *__restrict__ l_keep,
                                                     uint32_t *__restrict__ l_cnew, DevCounters *cnt, uint32_t hop_skip)
{
    HHGT_WAVE_PRIO();
    // The walk below is byte-serial per line; straight from global memory that is ~80 dependent loads per lane with
    // every wave of the grid resident at once, i.e. the kernel lasts one full latency chain (89 us for 136 k lines).
    // So the first 64 bytes of every line (CHROM .. FORMAT of a typical record) are fetched up front by four
    // independent 16-byte loads per lane and parked in LDS (row stride 17 dwords: conflict-free byte reads);
    // rd() serves the walk from there and falls back to global memory past byte 64.
    __shared__ uint32_t stage[256][17];
    const RegionFilter *rf = &rfv;   // by value in the kernel arguments: no upload, nothing to keep alive on the host
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_found = *d_nlines;
    const uint32_t n_lines = n_found < max_lines ? n_found : max_lines;
    if (i == 0) {
        cnt->n_lines = n_found;
        cnt->err_lines = n_found > max_lines ? (unsigned long long)(n_found - max_lines) : 0ull;
    }
    if (blockIdx.x * blockDim.x >= n_lines) {
        // the grid is sized by max_lines: the scans behind this kernel run over max_lines flags
        if (i < max_lines) l_keep[i] = l_cnew[i] = 0u;
        return;
    }
    uint32_t flags = 0;
    uint32_t s = 0, e = 0;
    bool staged = false;
    if (i < n_lines) {
        s = i ? nl[i - 1] + 1u : 0u;
        e = nl[i];
        staged = (uint64_t)s + 64ull <= n;
        if (staged) {
            u32x4_unaligned c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = *reinterpret_cast<const u32x4_unaligned *>(text + s + 16 * k);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                stage[threadIdx.x][4 * k + 0] = c[k].x;
                stage[threadIdx.x][4 * k + 1] = c[k].y;
                stage[threadIdx.x][4 * k + 2] = c[k].z;
                stage[threadIdx.x][4 * k + 3] = c[k].w;
            }
        }
    }
    __syncthreads();
    const uint8_t *mine = reinterpret_cast<const uint8_t *>(stage[threadIdx.x]);
    auto rd = [&](uint32_t x) -> uint32_t { return (staged && x - s < 64u) ? (uint32_t)mine[x - s] : (uint32_t)text[x]; };
    uint32_t soff = 0, pos0 = 0, refalt = 0, gtidx = 0, stride = 0;
    if (i < n_lines) {
        if (e > s && rd(e - 1u) == '\r') --e;  // bgzf_getline strips a trailing CR
        soff = e;
        if (e > s && rd(s) != '#') {
            flags = LF_RECORD;
            // walk the first 9 tab-separated fields
            uint32_t fs[10];
            uint32_t nf = 0;
            uint32_t p = s;
            fs[0] = s;
            while (p < e && nf < 9) {
                if (rd(p) == '\t') fs[++nf] = p + 1;
                ++p;
            }
            // nf = number of tabs found (<= 9); field k spans [fs[k], fs[k+1]-1) for k < nf
            bool bad = false;
            uint32_t fe[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) fe[k] = (uint32_t)k < nf ? fs[k + 1] - 1u : e;
            if (nf < 7 || (S > 0 && nf < 9)) bad = true;
            if (!bad) {
                // POS
                unsigned long long pos = 0;
                if (fe[1] == fs[1]) bad = true;
                for (uint32_t q = fs[1]; q < fe[1]; ++q) {
                    uint32_t c = rd(q);
                    if (c < '0' || c > '9') {
                        bad = true;
                        break;
                    }
                    pos = pos * 10ull + (c - '0');
                }
                bool in_region = true;
                if (!bad && rf->contig_len > 0) {
                    uint32_t cl = fe[0] - fs[0];
                    in_region = cl == (uint32_t)rf->contig_len;
                    for (uint32_t q = 0; in_region && q < cl; ++q)
                        in_region = rd(fs[0] + q) == (uint32_t)(uint8_t)rf->contig[q];
                    if (in_region && rf->has_range)
                        in_region = (long long)pos >= rf->beg && (long long)pos <= rf->end;
                }
                if (!bad) {
                    if (!in_region)
                        flags |= LF_DROP_REGION;
                    else {
                        // isSNP (cpp/vcfpp.h:990-1000): |REF| <= 1, n_allele <= 2, ALT in {A,C,G,T}
                        uint32_t reflen = fe[3] - fs[3], altlen = fe[4] - fs[4];
                        uint32_t a = altlen >= 1 ? rd(fs[4]) : 0;
                        bool snp = reflen == 1 && altlen == 1 && (a == 'A' || a == 'C' || a == 'G' || a == 'T');
                        if (rf->keep_multi && reflen == 1 && altlen > 1 && (altlen & 1u)) {
                            // non-reference mode: "B,B[,B...]" with every B in {A,C,G,T}
                            snp = true;
                            for (uint32_t q = 0; snp && q < altlen; ++q) {
                                const uint32_t ch = rd(fs[4] + q);
                                snp = (q & 1u) ? ch == ',' : (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
                            }
                        }
                        if (!snp)
                            flags |= LF_DROP_FILTER;
                        else {
                            pos0 = (uint32_t)(pos - 1ull);  // vcfpp.h:1118-1121
                            refalt = rd(fs[3]) | (a << 8);
                            if (S > 0) {
                                // GT key index inside FORMAT (bcf_get_genotypes looks the key up)
                                uint32_t q = fs[8], k = 0;
                                bool found = false;
                                while (q <= fe[8]) {
                                    uint32_t q2 = q;
                                    while (q2 < fe[8] && rd(q2) != ':') ++q2;
                                    if (q2 - q == 2 && rd(q) == 'G' && rd(q + 1u) == 'T') {
                                        found = true;
                                        break;
                                    }
                                    ++k;
                                    q = q2 + 1;
                                }
                                if (!found || k > 0xFFFu) bad = true;  // vcfpp.h:550-552 "genotypes not present" (or 4096 keys in front of it)
                                gtidx = k;
                                soff = fs[9];
                                if (!bad) {
                                    flags |= LF_KEEP;
                                    if (k == 0 && fe[8] - fs[8] == 2 && e - soff == 4u * S - 1u) flags |= LF_FAST;
                                    else if (k == 0 && (e - soff + 1u) % S == 0u) {
                                        // GT first and the sample columns S times some width of 5 .. 8 bytes ("a|b:dd\t"): the
                                        // bit-plane encoder tries them at that stride (round 4; every column is checked there)
                                        const uint32_t wd = (e - soff + 1u) / S;
                                        if (wd >= 5u && wd <= 8u) stride = wd;
                                    }
                                }
                            } else {
                                flags |= LF_KEEP;
                            }
                        }
                    }
                }
            }
            if (bad) flags = LF_RECORD | LF_MALFORMED;
            // CHROM run boundary: compare with the previous data line's CHROM
            bool cnew = true;
            if (i > 0) {
                uint32_t ps = i > 1 ? nl[i - 2] + 1u : 0u;
                uint32_t pe = nl[i - 1];
                // the previous line sits in the neighbouring thread's LDS row (same staging rule) unless this is thread 0
                const bool pstaged = threadIdx.x > 0u && (uint64_t)ps + 64ull <= n;
                const uint8_t *prev = reinterpret_cast<const uint8_t *>(stage[threadIdx.x > 0u ? threadIdx.x - 1u : 0u]);
                auto rdp = [&](uint32_t x) -> uint32_t { return (pstaged && x - ps < 64u) ? (uint32_t)prev[x - ps] : (uint32_t)text[x]; };
                if (pe > ps && rdp(ps) != '#') {
                    uint32_t q = 0;
                    cnew = false;
                    for (;;) {
                        uint32_t a = fs[0] + q < e ? rd(fs[0] + q) : '\t';
                        uint32_t b = ps + q < pe ? rdp(ps + q) : '\t';
                        if (a != b) {
                            cnew = true;
                            break;
                        }
                        if (a == '\t') break;
                        ++q;
                    }
                }
            }
            if (cnew) flags |= LF_CHROM_NEW;
        }
    }
    if (hop_skip) {
        // The hopping index (k_index_hop) never looked at the first hop_skip bytes behind a record's start: a record shorter
        // than that — fewer sample columns than the header declares, a parse error in htslib — has its newline in there and
        // arrives merged with its successor.  A KEPT record is read byte by byte by the encoders, which report the newline
        // (HHGT_ERR_MALFORMED); a record the region / isSNP filter DROPS is read by nobody, and the valid record merged into
        // it would vanish unreported.  So the dropped records' unexamined bytes are examined here, a wave per record
        // (rare: nothing is dropped from a 1000G-style SNP file), and a newline in there makes the record malformed too.
        const uint32_t lane = threadIdx.x & 63u;
        const bool chk = (flags & LF_RECORD) && !(flags & (LF_KEEP | LF_MALFORMED));
        unsigned long long todo = __ballot(chk);
        while (todo != 0ull) {   // (wave-uniform)
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t ss = (uint32_t)__builtin_amdgcn_readlane((int)s, l), ee = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
            const uint32_t hi = ee - ss < hop_skip ? ee : ss + hop_skip;   // (the index searched from (ss + hop_skip) & ~15 on)
            uint32_t hit = 0;
            // (eight KiB per step, the loads in flight together: behind the walk a dropped record is examined from end to end,
            // 20 KB at 5000 samples — one load per step made config 4's fixed-column stage 1.4 ms instead of 0.8)
            for (uint32_t x0 = ss + 16u * lane; x0 < hi; x0 += 8192u) {
                u32x4_t tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t x = x0 + 1024u * (uint32_t)u;
                    tv[u] = u32x4_t{0u, 0u, 0u, 0u};
                    if (x < hi && (uint64_t)x + 16ull <= n) tv[u] = *reinterpret_cast<const u32x4_unaligned *>(text + x);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t x = x0 + 1024u * (uint32_t)u;
                    if (x >= hi) continue;
                    uint32_t d[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
                    if ((uint64_t)x + 16ull > n)
                        for (uint32_t k = 0; k < 16u && (uint64_t)x + k < n; ++k) d[k >> 2] |= (uint32_t)text[x + k] << (8u * (k & 3u));
                    const uint32_t m = nl_mask4(d[0]) | (nl_mask4(d[1]) << 4) | (nl_mask4(d[2]) << 8) | (nl_mask4(d[3]) << 12);
                    hit |= hi - x >= 16u ? m : (m & ((1u << (hi - x)) - 1u));
                }
            }
            if (__ballot(hit != 0u) != 0ull && (int)lane == l) flags = (flags & LF_CHROM_NEW) | LF_RECORD | LF_MALFORMED;
        }
    }
    if (i < n_lines) {
        l_soff[i] = soff;
        l_lend[i] = e;
        l_pos[i] = pos0;
        l_refalt[i] = refalt | ((gtidx & 0xFFFu) << 16) | (stride << 28);   // (12 bits of GT key index: a FORMAT column of 4096 keys is reported, below)
        l_flags[i] = flags;
        l_keep[i] = (flags & LF_KEEP) ? 1u : 0u;
        l_cnew[i] = (flags & LF_CHROM_NEW) ? 1u : 0u;
    } else if (i < max_lines) {
        l_keep[i] = l_cnew[i] = 0u;
    }
    // statistics: one atomic per WORKGROUP per category.  (One per wave was 4 k atomic adds on one address for a chr1-sized
    // block — they retire one every ~12 ns, so the kernel took 66 us where its parse needs 20: round 4, from the kernel trace.)
    __shared__ uint32_t s_stat[4];
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 4u) s_stat[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t kinds[4] = {LF_RECORD, LF_DROP_REGION, LF_DROP_FILTER, LF_MALFORMED};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long b = __ballot(flags & kinds[k]);
        if (lane == 0 && b) atomicAdd(&s_stat[k], (uint32_t)__popcll(b));
    }
    __syncthreads();
    if (threadIdx.x < 4u && s_stat[threadIdx.x]) {
        unsigned long long *dst = threadIdx.x == 0u ? &cnt->n_records : (threadIdx.x == 1u ? &cnt->n_drop_region : (threadIdx.x == 2u ? &cnt->n_drop_filter : &cnt->n_malformed));
        atomicAdd(dst, (unsigned long long)s_stat[threadIdx.x]);
    }
}

// one thread per line; kept lines scatter to their compacted slot
__global__ __launch_bounds__(256) void k_compact_kept(
    const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ nl, const uint32_t *__restrict__ d_nlines,
    uint32_t max_lines, const uint32_t *__restrict__ l_soff,
    const uint32_t *__restrict__ l_lend, const uint32_t *__restrict__ l_pos, const uint32_t *__restrict__ l_refalt,
    const uint32_t *__restrict__ l_flags, const uint32_t *__restrict__ l_kidx, const uint32_t *__restrict__ l_crun,
    uint32_t *__restrict__ k_soff, uint32_t *__restrict__ k_lend, uint32_t *__restrict__ k_meta,
    uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_flag, uint64_t *__restrict__ run_first, uint8_t *__restrict__ run_names,
    uint32_t max_runs, const uint64_t *__restrict__ d_cursor, uint64_t v_capacity, uint32_t ring, uint32_t *__restrict__ d_start,
    uint32_t *__restrict__ d_stop, uint8_t *__restrict__ d_ref, uint8_t *__restrict__ d_alt, DevCounters *cnt, uint32_t strided)
{
    HHGT_WAVE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_found = *d_nlines;
    const uint32_t n_lines = n_found < max_lines ? n_found : max_lines;
    if (i >= n_lines) return;
    const uint64_t v_base = *d_cursor;
    const uint32_t flags = l_flags[i];
    if (flags & LF_CHROM_NEW) {
        // run id = number of CHROM_NEW flags before this line; the run's first kept index is the
        // exclusive kept-prefix here; its name is the CHROM field at the start of this line (copied out now: the text
        // buffer may be recycled before the host looks at the runs)
        uint32_t rid = l_crun[i];
        if (rid < max_runs) {
            run_first[rid] = l_kidx[i];
            const uint32_t s0 = i ? nl[i - 1] + 1u : 0u;
            uint8_t *nm = run_names + (size_t)rid * 32u;
            uint32_t q = 0;
            for (; q < 31u && (uint64_t)s0 + q < n; ++q) {
                const uint8_t ch = text[s0 + q];
                if (ch == '\t' || ch == '\n') break;
                nm[q] = ch;
            }
            for (; q < 32u; ++q) nm[q] = 0;
        }
    }
    if (i == n_lines - 1) {
        cnt->n_kept = (unsigned long long)l_kidx[n_lines];
        cnt->n_chrom_runs = (unsigned long long)l_crun[n_lines];
    }
    // kept lines that are not of the fixed-width shape go on the variable-width kernel's list: one atomic add per WAVE (config
    // 4 has 88 k such lines per pass — one returning add each on one address was most of this kernel's time)
    // (k_meta: bit 2 FAST, bits 8-19 GT key index, bits 20-23 column stride of a GT-first record whose columns are all of one
    // width; with `strided` — the bit-plane encoder runs — such a record is the tile kernel's, not the variable-width kernel's)
    const bool keep = (flags & LF_KEEP) != 0u;
    const bool slow = keep && !(flags & LF_FAST) && !(strided && (l_refalt[i < n_lines ? i : 0u] >> 28) != 0u);
    const unsigned long long sm = __ballot(slow);
    unsigned long long slot0 = 0ull;
    if (sm != 0ull) {
        const uint32_t lane = threadIdx.x & 63u;
        const int leader = __builtin_ctzll(sm);
        unsigned long long base = 0ull;
        if ((int)lane == leader) base = atomicAdd(&cnt->n_general, (unsigned long long)__popcll(sm));
        base = (unsigned long long)__shfl((long long)base, leader, 64);
        slot0 = base + (unsigned long long)__popcll(sm & ((1ull << lane) - 1ull));
    }
    if (!keep) return;
    const uint32_t k = l_kidx[i];
    uint64_t v = v_base + k;
    const uint32_t ra = l_refalt[i];
    k_soff[k] = l_soff[i];
    k_lend[k] = l_lend[i];
    k_meta[k] = (flags & LF_FAST) | ((ra >> 16) << 8);  // bit 2 = FAST, bits 8-19 = GT key index, bits 20-23 = column stride
    redo_flag[k] = 0u;                                  // (the tile kernel's "line already queued for the general path" mark)
    if (slow) redo_list[slot0] = k;
    if (ring) v %= v_capacity;   // ring of chunk columns: the tables wrap with it
    if (v < v_capacity) {
        uint32_t pos0 = l_pos[i];
        if (d_start) d_start[v] = pos0;
        if (d_stop) d_stop[v] = pos0 + 1u;  // vcfpp.h:1124-1127 with |REF| == 1
        if (d_ref) d_ref[v] = (uint8_t)(ra & 0xFFu);
        if (d_alt) d_alt[v] = (uint8_t)((ra >> 8) & 0xFFu);
    }
}

// ---------------------------------------------------------------------------------------------
// min_line: no line of this text can be shorter (0 = unknown).  Long lines: the hopping / walking index reads the text
// behind the bound only (-> the bytes it skips behind every record start; 0: the plain scan).
// mode (hhgt_set_index_mode; < 0: HHGT_INDEX_MODE, default 2): 0 the plain scan, 1 the hop by the bound, 2 the walk.
int index_mode_default()
{
    static const int m = [] {
        if (const char *e = getenv("HHGT_INDEX_MODE")) return atoi(e);
        if (const char *e = getenv("HHGT_INDEX_HOP")) return atoi(e) ? 2 : 0;   // (round 2's switch)
        return 2;
    }();
    return m < 0 ? 0 : (m > 2 ? 2 : m);
}

static uint32_t index_hop_skip(uint32_t min_line, int mode)
{
    if (mode < 0) mode = index_mode_default();
    return mode && min_line >= 1536u ? min_line - 16u : 0u;   // margin: the loads start at the 16-byte line in front of the target
}

int launch_index_newlines(const uint8_t *d_text, uint64_t n, uint32_t *d_slots, uint32_t *d_counts,
                          uint32_t n_regions, uint32_t min_line, uint32_t S, int mode, DevCounters *d_cnt, hipStream_t st)
{
    if (mode < 0) mode = index_mode_default();
    if (const uint32_t skip = index_hop_skip(min_line, mode)) {
        // regions per wave.  Hop, measured on the bench (3 M x 2504, index stage per step): 3.9 ms for 4 .. 7, 4.0-4.1 for 8, 4.3 for 12,
        // 4.4 for 16 and for "as many as make all waves of the launch resident at once" (11 on chr1) — longer walks per wave
        // cost more than a second, part-filled round of waves; fewer than 4 and the plain scan up to a range's first newline
        // (half a line per wave) starts to show
        static const int hop_k_env = getenv("HHGT_INDEX_HOP_K") ? atoi(getenv("HHGT_INDEX_HOP_K")) : 0;   // development
        const bool walk = mode >= 2 && S > 0;
        // The walk costs one dependent 1 KiB load per line, so a wave is a latency chain and every wave pays a plain scan up
        // to the first newline of its range (half a line, in U KiB windows): as few waves as still fill the chip once —
        // 8 workgroups of 4 waves on each of 256 CUs — at least the hop's 6 regions, at most 64 (the per-lane counters).
        const uint32_t k_walk = (n_regions + 8191u) / 8192u;
        uint32_t K = hop_k_env > 0 ? (uint32_t)hop_k_env : (walk ? (k_walk < 6u ? 6u : k_walk) : 6u);
        K = K > 64u ? 64u : K;
#define HOP_LAUNCH(U, W)                                                                                                      \
    hipLaunchKernelGGL((k_index_hop<U, W>), dim3(((n_regions + K - 1) / K + 3) / 4), dim3(256), 0, st, d_text, n, d_slots, d_counts, \
                       n_regions, skip, S, d_cnt, K)
        if (min_line >= 4096u) {
            if (walk) HOP_LAUNCH(5, true);
            else HOP_LAUNCH(5, false);
        } else {
            if (walk) HOP_LAUNCH(3, true);
            else HOP_LAUNCH(3, false);
        }
#undef HOP_LAUNCH
        HIP_TRY(hipGetLastError());
        return HHGT_OK;
    }
    hipLaunchKernelGGL(k_index_newlines, dim3((n_regions + 3) / 4), dim3(256), 0, st, d_text, n, d_slots,
                       d_counts, n_regions, d_cnt);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_compact_newlines(const uint32_t *d_slots, const uint32_t *d_counts, const uint32_t *d_prefix,
                            uint32_t n_regions, uint32_t *d_nl, uint32_t max_lines, hipStream_t st)
{
    hipLaunchKernelGGL(k_compact_newlines, dim3((n_regions + 255) / 256), dim3(256), 0, st, d_slots, d_counts,
                       d_prefix, n_regions, d_nl, max_lines);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_parse_fixed(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl, const uint32_t *d_nlines,
                       uint32_t max_lines, const RegionFilter &region, uint32_t S, uint32_t *l_soff, uint32_t *l_lend,
                       uint32_t *l_pos, uint32_t *l_refalt, uint32_t *l_flags, uint32_t *l_keep,
                       uint32_t *l_cnew, int mode, DevCounters *d_cnt, hipStream_t st)
{
    if (max_lines == 0) return HHGT_OK;
    if (mode < 0) mode = index_mode_default();
    // (the skip the hopping index used on this text, 0 for the plain scan: the same rule as launch_index_newlines; the walk
    // may have looked at nothing but the head of a record: a dropped record is examined from end to end)
    uint32_t hop_skip = index_hop_skip(S ? 2u * S + 17u : 0u, mode);
    if (hop_skip && mode >= 2) hop_skip = 0xFFFFFFFFu;
    hipLaunchKernelGGL(k_parse_fixed, dim3((max_lines + 255) / 256), dim3(256), 0, st, d_text, n, d_nl, d_nlines,
                       max_lines, region, S, l_soff, l_lend, l_pos, l_refalt, l_flags, l_keep, l_cnew, d_cnt, hop_skip);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

int launch_compact_kept(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl, const uint32_t *d_nlines, uint32_t max_lines,
                        const uint32_t *l_soff, const uint32_t *l_lend,
                        const uint32_t *l_pos, const uint32_t *l_refalt, const uint32_t *l_flags,
                        const uint32_t *l_kidx, const uint32_t *l_crun, uint32_t *k_soff, uint32_t *k_lend,
                        uint32_t *k_meta, uint32_t *redo_list, uint32_t *redo_flag, uint64_t *run_first, uint8_t *run_names,
                        uint32_t max_runs, const uint64_t *d_cursor, uint64_t v_capacity, uint32_t ring, uint32_t *d_start,
                        uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, DevCounters *d_cnt, bool strided, hipStream_t st)
{
    if (max_lines == 0) return HHGT_OK;
    hipLaunchKernelGGL(k_compact_kept, dim3((max_lines + 255) / 256), dim3(256), 0, st, d_text, n, d_nl, d_nlines, max_lines,
                       l_soff, l_lend, l_pos, l_refalt, l_flags, l_kidx, l_crun, k_soff, k_lend, k_meta, redo_list, redo_flag,
                       run_first, run_names, max_runs, d_cursor, v_capacity, ring, d_start, d_stop, d_ref, d_alt, d_cnt, strided ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
