// gapenc_ref.c — CPU statement of the bit-plane LZ4 encoder (csrc/lz4bits.hip, k_lz4_bitplanes), decision for
// decision, so that the kernel's streams can be compared byte for byte (tests/test_gpu_lz4_bitplanes.py) and its
// ratio studied without a GPU (tools/sim/gapenc.c is the exploration this came out of).  Not part of the product and
// not an oracle of the reference: whether a stream is VALID is checked by decoding it (liblz4 / oracle decoder).
//
// Input: a plane of n = 4096 bytes, all 0 or 1 — or, for the exception-aware instantiation of the kernel (planes with
// missing calls, config 4), 0, 1 or 0xF7 (-9).  Walks the NONZERO bytes ("ones") q_0 < q_1 < ... (a virtual one at -1 in
// front and one at n behind).  P[j+1] = q_j + 1; cls[j+1] = 1 if byte q_j is 0xF7.  A match copies bytes, so two ones
// only agree if their classes do: the table key carries the class of the one, a candidate of the other class is passed
// over, and a run of agreeing gaps ends in front of the first pair of ones whose classes differ (the zeros of that gap
// still agree).  On 0/1 planes every class is 0 and nothing changes.  For every one j:
//   * key = min(distance to the next one, 40), a 64-entry table of one indices; the table is updated one by one in
//     stream order (on the GPU: one ds_wrxchg_rtn_b32 per 64 ones — the LDS serves equal addresses in lane order);
//     what an insertion replaces is remembered (chain), so `depth` candidates can be tried per one: the table's entry,
//     what that one replaced, ... — the candidate that saves most wins (depth 0: no hash matches, runs only)
//   * candidates jc = previous ones with that key, most recent first.  The PICK among them is made on the gap bytes
//     (gaps clipped to 255, a 255 agrees with nothing): while the gaps behind the two ones are equal (at most 3 of
//     them) take gap + 1, then 1 + the smaller gap, plus the zeros the two have in common IN FRONT (at most 64: a match
//     is pulled back over them); the longest wins, ties go to the nearer one.  Only the winner is
//     evaluated exactly: its agreement may go on (to 16 gaps), costR = what coding the covered ones as literals
//     (+ offset-1 runs for gaps >= 5) would take; tailz = zeros left of the last covered one's gap
//   * pulled back over nb = min(64, zeros in front of q, zeros in front of the candidate) zeros
//   * taken (hv) iff cheaper than costR, forward part >= 4, total >= 6, q <= mflimit
//   * E = end of what the one codes (match end, else q + 1); nxt = first one at or behind E
// The parse follows nxt from the virtual one.  Every visited one emits its match M (start clamped to the end of the
// previous sequence) and a tail run T (offset 1) over the zeros from E (+1 if byte E-1 is a one, or E = 0) to the
// next one — if there are at least 4 of them, and at least 4 that the NEXT coded one's match does not pull back over
// anyway (a run costs 3 bytes; round 3: with the run rule blind to the pull-back and the pull-back held to 8 zeros the
// same candidates packed 8 % looser).
#include <stdint.h>
#include <string.h>

#define N_MAXONES 636
#define N_MAXONES_EXC 540   /* planes with a missing call: the exception-aware instantiation's list is shorter (csrc/lz4bits.hip BP_MAXONES_EXC) */
#define HLOG 6
#define GAPCLIP 40
#define MINM 6
#define BACK 64
#define TMIN 4     /* zeros an offset-1 run must cover (MINM: the total length a hash match must have) */
#define STEPS 16   /* exact extension of the chosen candidate */
#define PICK 3     /* full gaps the pick looks at */

static int put_len(uint8_t *out, int op, int r)
{
    while (r >= 255) { out[op++] = 255; r -= 255; }
    out[op++] = (uint8_t)r;
    return op;
}

static int put_seq(const uint8_t *in, uint8_t *out, int op, int anchor, int start, int len, int off)
{
    const int ll = start - anchor, ml = len - 4;
    out[op++] = (uint8_t)(((ll < 15 ? ll : 15) << 4) | (ml < 15 ? ml : 15));
    if (ll >= 15) op = put_len(out, op, ll - 15);
    memcpy(out + op, in + anchor, (size_t)ll);
    op += ll;
    out[op++] = (uint8_t)(off & 0xFF);
    out[op++] = (uint8_t)(off >> 8);
    if (ml >= 15) op = put_len(out, op, ml - 15);
    return op;
}

// returns the compressed size, or -1 if the plane is not handled by the bit-plane path (a byte > 1, too many ones)
// depth: candidates per one (low byte); bit 8 (0x100): with the lazy rule below
#define LAZY_MIN 3
int gapenc_ref(const uint8_t *in, int n, uint8_t *out, int depth)
{
    const int lazy = (depth >> 8) & 1;
    depth &= 0xFF;
    static __thread int chain[4100];
    static __thread int P[4096 + 4];
    static __thread uint8_t cls[4096 + 32];
    int m = 0, has_exc = 0;
    P[0] = 0;
    memset(cls, 0, sizeof(cls));
    for (int i = 0; i < n; ++i) {
        if (in[i] > 1 && in[i] != 0xF7) return -1;
        if (in[i]) {
            if (m >= N_MAXONES) return -1;
            cls[1 + m] = in[i] == 0xF7;
            has_exc |= in[i] == 0xF7;
            P[1 + m++] = i + 1;
        }
    }
    if (has_exc && m > N_MAXONES_EXC) return -1;
    P[m + 1] = P[m + 2] = n + 1;
    // gap bytes: GB[i] = zeros behind the one with P-index i (i = 0: the virtual one in front), clipped to 255; the last
    // real one and everything behind it read 255 = "agrees with nothing"
    static __thread uint8_t GB[4100 + 8];
    for (int i = 0; i < m; ++i) {
        const int g = P[i + 1] - P[i] - 1;
        GB[i] = (uint8_t)(g < 255 ? g : 255);
    }
    for (int i = m; i < m + 8; ++i) GB[i] = 255;
    uint32_t tab[1 << HLOG];
    memset(tab, 0, sizeof(tab));
    const int mflimit = n - 12, matchlimit = n - 5;
    static __thread int E[4100], nxt[4100], ms[4100], hv_[4100], off_[4100];
    for (int j = -1; j < m; ++j) {
        const int q = P[j + 1] - 1;
        int hv = 0, len = 0, nb = 0, c = 0;
        if (depth > 0 && j >= 0 && q + 12 <= n) {
            const uint32_t g1 = (uint32_t)(P[j + 2] - P[j + 1]);
            const uint32_t idx = (g1 < GAPCLIP ? g1 : GAPCLIP) ^ (cls[j + 1] ? 63u : 0u);
            int jc = (int)tab[idx] - 1;
            tab[idx] = (uint32_t)(j + 1);
            chain[j] = jc;
            // ---- pick: the candidate with the longest forward agreement, judged on the gap BYTES (gaps clipped to 255;
            // 255 never agrees): up to 3 equal gaps, then 1 + the smaller of the next pair.  Ties go to the nearer one.
            int best_score = -1, bj = -1, bk = 0;
            for (int dpt = 0; dpt < depth && jc >= 0; ++dpt, jc = chain[jc]) {
                int k = 0, score = 0;
                if (cls[jc + 1] != cls[j + 1]) continue;   // the first byte would differ
                while (k < PICK && GB[j + 1 + k] == GB[jc + 1 + k] && GB[j + 1 + k] != 255 && cls[j + 2 + k] == cls[jc + 2 + k]) {
                    score += GB[j + 1 + k] + 1;
                    ++k;
                }
                const int za = GB[j + 1 + k], zb = GB[jc + 1 + k];
                score += 1 + (za < zb ? za : zb);
                {   // the zeros in front that a match from this candidate would take along
                    const int fa = GB[j], fb = GB[jc];
                    const int bb = fa < fb ? fa : fb;
                    score += bb < BACK ? bb : BACK;
                }
                if (score > best_score) { best_score = score; bj = jc; bk = k; }
            }
            // ---- the chosen one, exactly: agreement continues past the third gap (rarely: periodic planes), then the
            // costs decide between this match and literals
            if (bj >= 0) {
                jc = bj;
                const int cc = P[jc + 1] - 1;
                int a = j + bk, b = jc + bk, clen = P[a + 1] - P[j + 1], costR = 0, tailz = 0;
                for (int s = 0; s < bk; ++s) {
                    const int g = GB[j + 1 + s];
                    costR += 1 + (g >= TMIN + 1 ? 4 : g);
                }
                for (int s = bk;; ++s) {
                    const int ga = P[a + 2] - P[a + 1] - 1, gb = P[b + 2] - P[b + 1] - 1;
                    costR += 1 + (ga >= TMIN + 1 ? 4 : ga);
                    if (s < PICK || ga != gb || ga >= 255 || a + 1 >= m || s >= STEPS || cls[a + 2] != cls[b + 2]) {   // s < PICK: the pick's own verdict
                        const int z = ga < gb ? ga : gb;
                        clen += 1 + z;
                        tailz = ga - z;
                        break;
                    }
                    clen += 1 + ga;
                    ++a; ++b;
                }
                const int gq = q + 1 - P[j] - 1, gc = cc + 1 - P[jc] - 1;
                int cnb = gq < gc ? gq : gc;
                if (cnb > BACK) cnb = BACK;
                const int costH = 3 + (clen + cnb >= 19 ? 1 : 0) - cnb + (tailz >= TMIN ? 3 : tailz);
                int end = q + clen;
                if (end > matchlimit) end = matchlimit;
                const int gain = costR - costH;
                if (gain > 0 && end - q >= 4 && end - (q - cnb) >= MINM && q <= mflimit) { hv = 1; len = end - q; nb = cnb; c = cc; }
            }
        }
        const int e = hv ? q + len : q + 1;
        E[j + 1] = e; ms[j + 1] = q - nb; hv_[j + 1] = hv; off_[j + 1] = q - c;
        int t = j + 1;
        while (t < m && P[t + 1] - 1 < e) ++t;
        nxt[j + 1] = t;
    }
    if (lazy) {
        // ---- one-step lazy rule (round 4; clevel 9, the file-writing paths' effort level): a one gives up its match when the
        // NEXT one lies inside that match and has a match of its own that ends further by at least LAZY_MIN + what giving up
        // costs — the zeros this one's match was pulled back over (they become literals) and the zeros between this one and
        // the start of the next one's match — it is then coded as a literal and the parse goes on with the next one.  Decided
        // per window of 64 ones (the kernel's windows: P-indices 64 w .. 64 w + 63; the successor must sit in the same window)
        // from the matches as the candidates left them; in a run of consecutive ones that all would give up, every other one
        // does, counted from the far end of the run (the last one of the run yields to a match that stays; the one in front of
        // it then keeps its own).  On bench-like planes: 6.52 -> 6.65 at depth 2, 6.95 -> 7.03 at depth 12 (liblz4 HC level
        // 5: 7.01).
        static __thread uint8_t L[4100];
        for (int i = 0; i <= m; ++i) {
            const int lane = i & 63;
            L[i] = 0;
            if (lane == 63 || i + 1 > m || !hv_[i] || !hv_[i + 1] || P[i + 1] - 1 >= E[i]) continue;
            const int nbi = (P[i] - 1) - ms[i];
            int between = ms[i + 1] - P[i];   // zeros between the one and the start of the next one's match
            if (between < 0) between = 0;
            L[i] = E[i + 1] - E[i] >= LAZY_MIN + nbi + between;
        }
        for (int i = 0; i <= m; ++i) {
            if (!L[i]) continue;
            int t = 0;
            while ((i & 63) + t < 64 && i + t <= m && L[i + t]) ++t;
            if (t & 1) hv_[i] = 2;   // (marked; applied below so that the run lengths see the untouched flags)
        }
        for (int i = 0; i <= m; ++i)
            if (hv_[i] == 2) {
                const int q = P[i] - 1;
                hv_[i] = 0;
                E[i] = q + 1;
                ms[i] = q;
                nxt[i] = i;
            }
    }
    int op = 0, prev_end = 0;
    for (int j = -1; j < m; j = nxt[j + 1]) {
        const int e = E[j + 1];
        if (hv_[j + 1]) {
            int st = ms[j + 1];
            if (st < prev_end) st = prev_end;
            op = put_seq(in, out, op, prev_end, st, e - st, off_[j + 1]);
            prev_end = e;
        }
        const int rs = e + ((e == 0 || in[e - 1]) ? 1 : 0);
        int re = P[nxt[j + 1] + 1] - 1;
        if (re > matchlimit) re = matchlimit;
        const int zt = re - rs, kn = nxt[j + 1];
        int nbk = 0;   // zeros of this run that the next coded one's match starts in front of
        if (kn < m && hv_[kn + 1]) {
            nbk = (P[kn + 1] - 1) - ms[kn + 1];
            if (nbk > zt) nbk = zt;
        }
        if (zt >= TMIN && zt - nbk >= TMIN && rs <= mflimit) {
            op = put_seq(in, out, op, prev_end, rs, re - rs, 1);
            prev_end = re;
        }
    }
    const int ll = n - prev_end;
    out[op++] = (uint8_t)((ll < 15 ? ll : 15) << 4);
    if (ll >= 15) op = put_len(out, op, ll - 15);
    memcpy(out + op, in + prev_end, (size_t)ll);
    op += ll;
    return op;
}
