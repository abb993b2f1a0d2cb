// gapenc.c — CPU model of the "list of ones" LZ4 encoder for 0/1 byte planes (development tool; the GPU kernel
// k_lz4_bitplanes in csrc/lz4bits.hip follows it step by step, so its output can be compared byte for byte).
//
// A 4 KiB plane of a genotype matrix is mostly zeros: ~260 ones.  Instead of looking at every byte position, the
// encoder walks the ONES.  With q_0 < q_1 < ... the positions of the ones and g_j = q_{j+1} - q_j - 1 the zeros
// behind one j, every one has two ways to be coded:
//   H  a match that starts at the one (pulled back over up to 8 literal zeros in front of it) and copies from an
//      earlier one with the same 12-bit context (hash table keyed on the bits q..q+11, most recent one wins);
//      its length follows from comparing GAPS, not bytes: equal gaps, then the shorter of the first unequal pair
//   R  the one as a literal
// and behind whatever ends at E (H: the match's end, R: q + 1) the zeros up to the next one go out as an offset-1
// run if at least 6 of them can (the first zero behind a one has to be a literal: its predecessor is a 1).
// Which ones are coded at all is a linked list: nxt(j) = first one at or behind E_j — static per one — so the greedy
// parse is pointer chasing, done per window of 64 ones (hash table updates are window-granular: a one sees the
// ones of earlier windows only).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINM 6
static int g_costrule = 0, g_pullmode = 0;
#define BACK 8
#define HLOG 10

typedef struct { int start, len, off; } Seq;

static int emit(const uint8_t *in, int n, const Seq *sq, int ns, uint8_t *out)
{
    int op = 0, anchor = 0;
    for (int k = 0; k < ns; ++k) {
        int ll = sq[k].start - anchor, ml = sq[k].len - 4;
        out[op++] = (uint8_t)(((ll < 15 ? ll : 15) << 4) | (ml < 15 ? ml : 15));
        if (ll >= 15) { int r = ll - 15; while (r >= 255) { out[op++] = 255; r -= 255; } out[op++] = (uint8_t)r; }
        memcpy(out + op, in + anchor, ll); op += ll;
        out[op++] = sq[k].off & 0xFF; out[op++] = sq[k].off >> 8;
        if (ml >= 15) { int r = ml - 15; while (r >= 255) { out[op++] = 255; r -= 255; } out[op++] = (uint8_t)r; }
        anchor = sq[k].start + sq[k].len;
    }
    int ll = n - anchor;
    out[op++] = (uint8_t)((ll < 15 ? ll : 15) << 4);
    if (ll >= 15) { int r = ll - 15; while (r >= 255) { out[op++] = 255; r -= 255; } out[op++] = (uint8_t)r; }
    memcpy(out + op, in + anchor, ll); op += ll;
    return op;
}

static int lz4_decode(const uint8_t *src, int n, uint8_t *dst, int cap)
{
    int ip = 0, op = 0;
    while (ip < n) {
        int tok = src[ip++], ll = tok >> 4;
        if (ll == 15) { int b; do { b = src[ip++]; ll += b; } while (b == 255); }
        if (op + ll > cap) return -1;
        memcpy(dst + op, src + ip, ll); ip += ll; op += ll;
        if (ip >= n) break;
        int off = src[ip] | (src[ip + 1] << 8); ip += 2;
        int ml = tok & 15;
        if (ml == 15) { int b; do { b = src[ip++]; ml += b; } while (b == 255); }
        ml += 4;
        if (off == 0 || off > op || op + ml > cap) return -2;
        for (int k = 0; k < ml; ++k) { dst[op] = dst[op - off]; ++op; }
    }
    return op;
}

// pos[-1] = -1 (virtual one in front of the stream); pos[m] = n (virtual one behind it)
static int gap_encode(const uint8_t *in, int n, uint8_t *out, int *nseq_out, int window, int intra)
{
    static int posbuf[4096 + 3];
    int *pos = posbuf + 1;
    int m = 0;
    pos[-1] = -1;
    for (int i = 0; i < n; ++i) if (in[i]) pos[m++] = i;
    pos[m] = n;
    pos[m + 1] = n;   // so that gap(m) reads as 0
    static uint16_t tab[1 << HLOG];
    memset(tab, 0, sizeof(tab));   // 0 = empty, else one index + 1 (index -1..m-1 -> 0..m)
    static Seq sq[8192];
    int ns = 0;
    const int mflimit = n - 12, matchlimit = n - 5;
    // per one: E (end of what it codes), nxt, the H match if any
    static int E[4097], nxt[4097], hs[4097], hl[4097], ho[4097];
    // the virtual one j = -1 is entry 0 of these arrays: index with j + 1
    for (int w0 = -1; w0 < m; w0 += window) {
        int w1 = w0 + window < m ? w0 + window : m;
        for (int j = w0; j < w1; ++j) {
            const int q = pos[j];
            int h = 0, nb = 0, c = 0, costR = 0, tailz = 0;
            if (j >= 0 && q + 12 <= n) {
                uint32_t ctx = 0;
                for (int k = 0; k < 12; ++k) ctx |= (uint32_t)(in[q + k] & 1) << k;
                const uint32_t idx = (ctx * 2654435761u) >> (32 - HLOG);
                int jc = (int)tab[idx] - 1;            // candidate one (index), -1 = none... (index -1 is never inserted)
                if (intra) {                            // exact: most recent earlier one with the same context
                    for (int jj = j - 1; jj >= w0 && jj >= 0; --jj) {
                        uint32_t c2 = 0;
                        if (pos[jj] + 12 > n) continue;
                        for (int k = 0; k < 12; ++k) c2 |= (uint32_t)(in[pos[jj] + k] & 1) << k;
                        if (((c2 * 2654435761u) >> (32 - HLOG)) == idx) { jc = jj; break; }
                    }
                }
                if (jc >= 0 && jc < j) {
                    c = pos[jc];
                    // forward length from the gaps: equal gaps, then 1 + the shorter of the first unequal pair
                    int a = j, b = jc, len = 0, steps = 0;
                    costR = 0;
                    for (;;) {
                        const int ga = pos[a + 1] - pos[a] - 1, gb = pos[b + 1] - pos[b] - 1;
                        if (ga != gb || a + 1 >= m || steps >= 8) {
                            const int z = ga < gb ? ga : gb;
                            len += 1 + z;
                            // the last one (partly) covered: what R pays for it, and what is left of its gap for H
                            costR += 1 + (ga >= MINM + 1 ? 4 : ga);
                            tailz = ga - z;
                            break;
                        }
                        len += 1 + ga;
                        costR += 1 + (ga >= MINM + 1 ? 4 : ga);
                        ++a; ++b; ++steps;
                    }
                    h = len;
                    // zeros in front: literal zeros of a short gap before q, zeros before the source
                    const int gq = q - pos[j - 1] - 1, gc = c - pos[jc - 1] - 1;
                    int zb = gq <= MINM ? gq : 0;     // (a longer gap goes out as a run up to q: nothing to pull)
                    if (g_pullmode) zb = gq;
                    nb = zb < gc ? zb : gc;
                    if (nb > BACK) nb = BACK;
                }
            }
            int e = q + 1, st = 0, ln = 0;
            // H against R on the same span: 3 bytes (+1 for a long match) minus the literals pulled in, plus the zeros left
            // behind it, against one literal per one and literal / literal + run per gap
            int costH = 3 + (h + nb >= 19 ? 1 : 0) - nb + (tailz >= MINM ? 3 : tailz);
            if (g_costrule && costH >= costR) h = 0, nb = 0;
            if (h + nb >= MINM && q - nb <= mflimit) {
                int end = q + h;
                if (end > matchlimit) end = matchlimit;
                if (end - (q - nb) >= MINM) { e = end; st = q - nb; ln = end - st; }
            }
            E[j + 1] = e; hs[j + 1] = st; hl[j + 1] = ln; ho[j + 1] = q - c;
            // first one at or behind e (e > q, so nxt > j)
            int t = j + 1;
            while (t < m && pos[t] < e) ++t;
            nxt[j + 1] = t;
        }
        // insert the window's ones (most recent wins)
        for (int j = w0 < 0 ? 0 : w0; j < w1; ++j) {
            if (pos[j] + 12 > n) continue;
            uint32_t ctx = 0;
            for (int k = 0; k < 12; ++k) ctx |= (uint32_t)(in[pos[j] + k] & 1) << k;
            tab[(ctx * 2654435761u) >> (32 - HLOG)] = (uint16_t)(j + 1);
        }
    }
    // walk the list
    int prev_end = 0;
    for (int j = -1; j < m; j = nxt[j + 1]) {
        if (hl[j + 1]) {
            int st = hs[j + 1], en = st + hl[j + 1];
            if (st < prev_end) st = prev_end;              // the previous sequence took some of the zeros in front
            if (en - st >= 4 && st <= mflimit) { sq[ns++] = (Seq){st, en - st, ho[j + 1]}; prev_end = en; }
            else if (hl[j + 1]) { /* too short after clamping: the bytes stay literals; E stays (its zeros follow) */ }
        }
        // zeros [E, next one): an offset-1 run from the first zero whose predecessor is a zero
        const int e = E[j + 1], qn = pos[nxt[j + 1]];
        int rs = e;
        if (e == 0 || in[e - 1]) rs = e + 1;               // first zero behind a one (or position 0) is a literal
        int re = qn;
        if (re > matchlimit) re = matchlimit;
        if (rs < prev_end) rs = prev_end;
        if (re - rs >= MINM && rs <= mflimit) { sq[ns++] = (Seq){rs, re - rs, 1}; prev_end = re; }
    }
    if (nseq_out) *nseq_out = ns;
    return emit(in, n, sq, ns, out);
}

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
static double u01(uint64_t seed, uint64_t stream, uint64_t i)
{
    uint64_t ctr = i * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + seed;
    return (double)(mix64(ctr) >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv)
{
    const int N = 4096, S = 2504, planes = argc > 1 ? atoi(argv[1]) : 400;
    static uint8_t plane[4096 + 64], out[8192], back[4096];
    for (int variant = 0; variant < 12; ++variant) {
        g_costrule = variant >= 4;
        g_pullmode = variant >= 8;
        const int window = variant % 4 == 0 ? 1 : 64, intra = variant % 4 == 2;
        long tot = 0, seqs = 0, ones = 0;
        for (int pl = 0; pl < planes; ++pl) {
            uint64_t seed = 1000 + pl % 22, v0 = (uint64_t)(pl / 22) * N;
            uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ull);
            uint64_t sh = (uint64_t)(pl * 7919 % (2 * S)) * 0x9E3779B97F4A7C15ull;
            for (int v = 0; v < N; ++v) {
                double lo = 1.0 / (2.0 * S), p = lo * pow(0.5 / lo, u01(seed, 3, v0 + v));
                if (variant % 4 == 3) p = p * 4 > 0.5 ? 0.5 : p * 4;   // a denser cohort
                uint64_t kv = key ^ ((v0 + v) * 0xD1B54A32D192ED03ull);
                plane[v] = (uint32_t)(mix64(kv ^ sh) >> 32) < (uint32_t)fmin(floor(p * 4294967296.0), 4294967295.0);
                ones += plane[v];
            }
            int ns;
            int c = gap_encode(plane, N, out, &ns, window, intra);
            int d = lz4_decode(out, c, back, N);
            if (d != N || memcmp(back, plane, N)) { printf("DECODE MISMATCH plane %d (d=%d)\n", pl, d); return 1; }
            tot += c < N ? c : N;
            seqs += ns;
        }
        printf("cost %d pull %d %-44s ratio %.3f  %.1f seq/plane  %.0f B  (%.0f ones)\n", g_costrule, g_pullmode,
               variant % 4 == 0 ? "table updated per one (exact recency)" : variant % 4 == 1 ? "window of 64 ones, no intra-window" :
               variant % 4 == 2 ? "window of 64 ones + exact intra-window" : "window 64, 4x denser planes",
               (double)planes * N / tot, (double)seqs / planes, (double)tot / planes, (double)ones / planes);
    }
    return 0;
}
