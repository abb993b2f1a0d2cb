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
// k_index_hop<U>: the same index for text whose lines cannot be shorter than `skip` bytes — a record with S sample
// columns has at least 2 S + 17 bytes in front of its newline, so behind every line start that many bytes need not be
// looked at.  For the 1000G shape (lines of 4 S + 58 bytes) that is half of the text: the pass reads ~5 KB per variant
// where k_index_newlines reads 10 KB.  A wave owns HOP_K (<= 64, the launcher's choice: 6) consecutive regions and walks
// them line by line:
// from `pos` it loads U KiB (16 B per lane, all loads in flight together), takes the first newline, appends it to the
// slot list of the region it lies in (same slots / counts layout as k_index_newlines: everything downstream is
// unchanged), looks at the first byte of the next line ('#' header lines and empty lines are short: no skip behind
// them) and hops.  The first newline of a wave's range is found by plain scanning, so ranges need no hand-over.
// A line that IS shorter than the bound (fewer sample columns than the header declares) is malformed either way; when
// its newline falls into a hop it merges with the next line, and the newline is found where the merged line is read:
// by the general encoder inside the sample columns of a KEPT record (encode.hip), by k_parse_fixed in the skipped bytes
// of a record the filters DROP (nobody else reads those) — the same error, one record later.
template <int U>
__global__ __launch_bounds__(256) void k_index_hop(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ slots,
                                                   uint32_t *__restrict__ counts, uint32_t n_regions, uint32_t skip,
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
    while (pos < endw) {
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
        bool found = false, nb_have = false;
        uint32_t nb_reg = 0;
        uint64_t p = 0;
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
        if (p >= endw) break;
        {
            const uint32_t rr = (uint32_t)(p / INDEX_REGION), local = rr - r_first;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)local);
            if (c < INDEX_CAP) {
                if (lane == 0) slots[(size_t)rr * INDEX_CAP + c] = (uint32_t)p;
            } else
                overflow = true;
            cntv += lane == local ? 1u : 0u;
        }
        uint32_t nb = 0x0Au;
        if (nb_have) nb = nb_reg;                 // the byte behind the newline came with the same 16 bytes (15 of 16 cases)
        else if (p + 1 < n) nb = text[p + 1];
        pos = p + 1 + ((nb == '#' || nb == 0x0Au) ? 0u : skip);
        // the terminator of the text's last line is always looked at: a file cut off in mid-line ends in a line shorter
        // than the bound, which has to reach the parser (and be reported), not vanish in a hop
        if (pos > last_term) pos = p + 1 > last_term ? p + 1 : last_term;
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
    uint32_t soff = 0, pos0 = 0, refalt = 0, gtidx = 0;
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
                                if (!found) bad = true;  // vcfpp.h:550-552 "genotypes not present"
                                gtidx = k;
                                soff = fs[9];
                                if (!bad) {
                                    flags |= LF_KEEP;
                                    if (k == 0 && fe[8] - fs[8] == 2 && e - soff == 4u * S - 1u) flags |= LF_FAST;
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
            for (uint32_t x = ss + 16u * lane; x < hi; x += 1024u) {
                uint32_t d[4] = {0u, 0u, 0u, 0u};
                if ((uint64_t)x + 16ull <= n) {
                    const u32x4_unaligned t = *reinterpret_cast<const u32x4_unaligned *>(text + x);
                    d[0] = t.x, d[1] = t.y, d[2] = t.z, d[3] = t.w;
                } else {
                    for (uint32_t k = 0; k < 16u && (uint64_t)x + k < n; ++k) d[k >> 2] |= (uint32_t)text[x + k] << (8u * (k & 3u));
                }
                const uint32_t m = nl_mask4(d[0]) | (nl_mask4(d[1]) << 4) | (nl_mask4(d[2]) << 8) | (nl_mask4(d[3]) << 12);
                hit |= hi - x >= 16u ? m : (m & ((1u << (hi - x)) - 1u));
            }
            if (__ballot(hit != 0u) != 0ull && (int)lane == l) flags = (flags & LF_CHROM_NEW) | LF_RECORD | LF_MALFORMED;
        }
    }
    if (i < n_lines) {
        l_soff[i] = soff;
        l_lend[i] = e;
        l_pos[i] = pos0;
        l_refalt[i] = refalt | (gtidx << 16);
        l_flags[i] = flags;
        l_keep[i] = (flags & LF_KEEP) ? 1u : 0u;
        l_cnew[i] = (flags & LF_CHROM_NEW) ? 1u : 0u;
    } else if (i < max_lines) {
        l_keep[i] = l_cnew[i] = 0u;
    }
    // statistics: one atomic per wave per category
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long b;
    b = __ballot(flags & LF_RECORD);
    if (lane == 0 && b) atomicAdd(&cnt->n_records, (unsigned long long)__popcll(b));
    b = __ballot(flags & LF_DROP_REGION);
    if (lane == 0 && b) atomicAdd(&cnt->n_drop_region, (unsigned long long)__popcll(b));
    b = __ballot(flags & LF_DROP_FILTER);
    if (lane == 0 && b) atomicAdd(&cnt->n_drop_filter, (unsigned long long)__popcll(b));
    b = __ballot(flags & LF_MALFORMED);
    if (lane == 0 && b) atomicAdd(&cnt->n_malformed, (unsigned long long)__popcll(b));
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
    uint32_t *__restrict__ d_stop, uint8_t *__restrict__ d_ref, uint8_t *__restrict__ d_alt, DevCounters *cnt)
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
    if (!(flags & LF_KEEP)) return;
    const uint32_t k = l_kidx[i];
    uint64_t v = v_base + k;
    const uint32_t ra = l_refalt[i];
    k_soff[k] = l_soff[i];
    k_lend[k] = l_lend[i];
    k_meta[k] = (flags & LF_FAST) | ((ra >> 16) << 8);  // bit2 = FAST, bits 8.. = GT key index
    redo_flag[k] = 0u;                                  // (the tile kernel's "line already queued for the general path" mark)
    if (!(flags & LF_FAST)) {
        unsigned long long slot = atomicAdd(&cnt->n_general, 1ull);
        redo_list[slot] = k;
    }
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
// min_line: no line of this text can be shorter (0 = unknown).  Long lines: the hopping index reads the text behind the
// bound only (-> the bytes it skips behind every record start; 0: the plain scan); HHGT_INDEX_HOP=0 keeps the plain scan.
static uint32_t index_hop_skip(uint32_t min_line)
{
    static const int hop = getenv("HHGT_INDEX_HOP") ? atoi(getenv("HHGT_INDEX_HOP")) : 1;
    return hop && min_line >= 1536u ? min_line - 16u : 0u;   // margin: the loads start at the 16-byte line in front of the target
}

int launch_index_newlines(const uint8_t *d_text, uint64_t n, uint32_t *d_slots, uint32_t *d_counts,
                          uint32_t n_regions, uint32_t min_line, DevCounters *d_cnt, hipStream_t st)
{
    if (const uint32_t skip = index_hop_skip(min_line)) {
        // regions per wave.  Measured on the bench (3 M x 2504, index stage per step): 3.9 ms for 4 .. 7, 4.0-4.1 for 8, 4.3 for 12,
        // 4.4 for 16 and for "as many as make all waves of the launch resident at once" (11 on chr1) — longer walks per wave
        // cost more than a second, part-filled round of waves; fewer than 4 and the plain scan up to a range's first newline
        // (half a line per wave) starts to show
        static const int hop_k_env = getenv("HHGT_INDEX_HOP_K") ? atoi(getenv("HHGT_INDEX_HOP_K")) : 0;   // development
        uint32_t K = hop_k_env > 0 ? (uint32_t)hop_k_env : 6u;
        K = K > 64u ? 64u : K;
#define HOP_LAUNCH(U)                                                                                                      \
    hipLaunchKernelGGL((k_index_hop<U>), dim3(((n_regions + K - 1) / K + 3) / 4), dim3(256), 0, st, d_text, n, d_slots, d_counts, \
                       n_regions, skip, d_cnt, K)
        if (min_line >= 4096u) HOP_LAUNCH(5);
        else HOP_LAUNCH(3);
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
                       uint32_t *l_cnew, DevCounters *d_cnt, hipStream_t st)
{
    if (max_lines == 0) return HHGT_OK;
    // (the skip the hopping index used on this text, 0 for the plain scan: the same rule as launch_index_newlines)
    const uint32_t hop_skip = index_hop_skip(S ? 2u * S + 17u : 0u);
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
                        uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, DevCounters *d_cnt, hipStream_t st)
{
    if (max_lines == 0) return HHGT_OK;
    hipLaunchKernelGGL(k_compact_kept, dim3((max_lines + 255) / 256), dim3(256), 0, st, d_text, n, d_nl, d_nlines, max_lines,
                       l_soff, l_lend, l_pos, l_refalt, l_flags, l_kidx, l_crun, k_soff, k_lend, k_meta, redo_list, redo_flag,
                       run_first, run_names, max_runs, d_cursor, v_capacity, ring, d_start, d_stop, d_ref, d_alt, d_cnt);
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}
