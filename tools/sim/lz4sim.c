// lz4sim.c — CPU simulation of candidate-selection / parse strategies on genotype byte planes (development tool:
// decides what the GPU encoder should look for before it is written).  Planes follow bench.py's generator
// (haplohyped_varawareml_amd/synth.py: allele frequency log-uniform on [1/(2S), 0.5] per variant, splitmix64 rule).
//   gcc -O2 -o /tmp/lz4sim tools/sim/lz4sim.c -lm && /tmp/lz4sim
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N 4096
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

// LZ4 cost of a sequence: token + literal-length extension + literals + offset + match-length extension
static int seq_cost(int ll, int ml)
{
    int c = 1 + ll + 2;
    if (ll >= 15) c += (ll - 15) / 255 + 1;
    if (ml - 4 >= 15) c += (ml - 4 - 15) / 255 + 1;
    return c;
}
static int last_cost(int ll) { return 1 + ll + (ll >= 15 ? (ll - 15) / 255 + 1 : 0); }

static int match_len(const uint8_t *b, int i, int c, int limit)
{
    int l = 0;
    while (i + l < limit && b[i + l] == b[c + l]) ++l;
    return l;
}

// generic greedy: cand(i) gives best (len, off) at i; returns compressed size
typedef struct { int len, off; } Cand;
typedef Cand (*CandFn)(const uint8_t *b, int i, void *st);
static int g_minmatch = 6;

static int greedy(const uint8_t *b, CandFn f, void *st, int *nseq)
{
    const int mflimit = N - 12, matchlimit = N - 5;
    int i = 0, anchor = 0, out = 0, ns = 0;
    while (i <= mflimit) {
        Cand c = f(b, i, st);
        if (c.len > matchlimit - i) c.len = matchlimit - i;
        if (c.len >= g_minmatch) {
            out += seq_cost(i - anchor, c.len);
            ++ns;
            // positions inside the match are still fed to the state (hash insertion etc.)
            for (int k = i + 1; k < i + c.len && k <= mflimit; ++k) f(b, -k - 1, st);   // negative: insert only
            i += c.len;
            anchor = i;
        } else
            ++i;
    }
    out += last_cost(N - anchor);
    if (nseq) *nseq = ns;
    return out < N ? out : N;
}

// ---- candidate generators -------------------------------------------------------------------------------------
static int run_len(const uint8_t *b, int i)
{
    if (i == 0) return 0;
    int l = 0;
    while (i + l < N - 5 && b[i + l] == b[i - 1]) ++l;
    return l;
}
static Cand cand_run(const uint8_t *b, int i, void *st)
{
    (void)st;
    Cand c = {0, 1};
    if (i < 0) return c;
    c.len = run_len(b, i);
    return c;
}

typedef struct { int tab[1 << 16]; int keybits; } HashSt;
static uint32_t key_at(const uint8_t *b, int i, int kb)
{
    uint32_t k = 0;
    for (int j = 0; j < kb; ++j) k |= (uint32_t)(b[i + j] & 1) << j;   // planes are 0/1 here
    return k;
}
// exact most recent previous occurrence of the same kb-bit context + run candidate, run-dominance rule
static Cand cand_hash(const uint8_t *b, int i, void *stv)
{
    HashSt *st = (HashSt *)stv;
    Cand c = {0, 1};
    int ins = i < 0 ? -i - 1 : i;
    uint32_t k = key_at(b, ins, st->keybits);
    int prev = st->tab[k];
    st->tab[k] = ins;
    if (i < 0) return c;
    int r = run_len(b, i);
    int h = prev >= 0 ? match_len(b, i, prev, N - 5) : 0;
    // dominance: a hash match that starts 1-2 bytes before a run and ends inside it is dropped
    if (h >= g_minmatch) {
        int r1 = i + 1 < N ? run_len(b, i + 1) : 0, r2 = i + 2 < N ? run_len(b, i + 2) : 0;
        if ((r1 >= g_minmatch && r1 + 1 > h) || (r2 >= g_minmatch && r2 + 2 > h)) h = 0;
    }
    if (r >= h) { c.len = r; c.off = 1; }
    else { c.len = h; c.off = i - prev; }
    return c;
}

// longest match among offsets 1..maxoff (bit-parallel multi-offset matcher), optionally also offsets k*64
typedef struct { int maxoff; int far; } OffSt;
static Cand cand_offsets(const uint8_t *b, int i, void *stv)
{
    OffSt *st = (OffSt *)stv;
    Cand c = {0, 1};
    if (i < 0) return c;
    for (int d = 1; d <= st->maxoff && d <= i; ++d) {
        int l = match_len(b, i, i - d, N - 5);
        if (l > c.len) { c.len = l; c.off = d; }
    }
    if (st->far)
        for (int d = 128; d <= i; d += 64) {
            int l = match_len(b, i, i - d, N - 5);
            if (l > c.len) { c.len = l; c.off = d; }
        }
    // same dominance rule against the run that starts 1-2 bytes later
    if (c.off != 1 && c.len >= g_minmatch) {
        int r1 = run_len(b, i + 1), r2 = run_len(b, i + 2);
        if ((r1 >= g_minmatch && r1 + 1 > c.len) || (r2 >= g_minmatch && r2 + 2 > c.len)) c.len = 0;
    }
    return c;
}

// hash lookups / insertions only at "edge" positions (mode 1: b[i] == 1; mode 2: b[i] != b[i-1]); everything else
// can only continue a run — what an encoder that walks the list of ones (not the bytes) would see
typedef struct { int tab[1 << 16]; int keybits, mode, backext; } EdgeSt;
static Cand cand_edge(const uint8_t *b, int i, void *stv)
{
    EdgeSt *st = (EdgeSt *)stv;
    Cand c = {0, 1};
    int ins = i < 0 ? -i - 1 : i;
    int edge = st->mode == 1 ? b[ins] == 1 : (ins == 0 || b[ins] != b[ins - 1]);
    int prev = -1;
    if (edge) {
        uint32_t k = key_at(b, ins, st->keybits);
        prev = st->tab[k];
        st->tab[k] = ins;
    }
    if (i < 0) return c;
    int r = run_len(b, i);
    int h = prev >= 0 ? match_len(b, i, prev, N - 5) : 0;
    if (!edge && st->backext) {
        // the next edge's candidate, pulled back over the zeros (equal bytes) in front of it
        int e = i + 1;
        while (e < N - 12 && e - i <= st->backext && !(st->mode == 1 ? b[e] == 1 : b[e] != b[e - 1])) ++e;
        if (e < N - 12 && e - i <= st->backext) {
            int cnd = st->tab[key_at(b, e, st->keybits)];
            if (cnd >= 0 && cnd - (e - i) >= 0 && cnd < e) {
                int ok = 1;
                for (int k = 1; k <= e - i; ++k) ok &= b[cnd - k] == b[e - k];
                if (ok) { prev = cnd - (e - i); h = match_len(b, i, prev, N - 5); }
            }
        }
    }
    if (h >= g_minmatch) {
        int r1 = run_len(b, i + 1), r2 = run_len(b, i + 2);
        if ((r1 >= g_minmatch && r1 + 1 > h) || (r2 >= g_minmatch && r2 + 2 > h)) h = 0;
    }
    if (r >= h) { c.len = r; c.off = 1; }
    else { c.len = h; c.off = i - prev; }
    return c;
}

// hash chain of depth D (LZ4HC-like), exact per-context chains
typedef struct { int head[1 << 16]; int prev[N]; int keybits, depth; } ChainSt;
static Cand cand_chain(const uint8_t *b, int i, void *stv)
{
    ChainSt *st = (ChainSt *)stv;
    Cand c = {0, 1};
    int ins = i < 0 ? -i - 1 : i;
    uint32_t k = key_at(b, ins, st->keybits);
    int p = st->head[k];
    st->prev[ins] = p;
    st->head[k] = ins;
    if (i < 0) return c;
    int r = run_len(b, i), h = 0, hp = -1;
    for (int d = 0; d < st->depth && p >= 0; ++d, p = st->prev[p]) {
        int l = match_len(b, i, p, N - 5);
        if (l > h) { h = l; hp = p; }
    }
    if (h >= g_minmatch) {
        int r1 = run_len(b, i + 1), r2 = run_len(b, i + 2);
        if ((r1 >= g_minmatch && r1 + 1 > h) || (r2 >= g_minmatch && r2 + 2 > h)) h = 0;
    }
    if (r >= h) { c.len = r; c.off = 1; }
    else { c.len = h; c.off = i - hp; }
    return c;
}

int main(int argc, char **argv)
{
    const int S = 2504, planes = argc > 1 ? atoi(argv[1]) : 400;
    static uint8_t plane[N + 64];
    static double p[N];
    long tot[32] = {0}, seqs[32] = {0};
    const char *names[32] = {0};
    for (int pl = 0; pl < planes; ++pl) {
        // one block = 4096 consecutive variants of one sample's haplotype
        uint64_t seed = 1000 + pl % 22, v0 = (uint64_t)(pl / 22) * N;
        for (int v = 0; v < N; ++v) {
            double lo = 1.0 / (2.0 * S);
            p[v] = lo * pow(0.5 / lo, u01(seed, 3, v0 + v));
        }
        uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ull);
        uint64_t sh = (uint64_t)(pl * 7919 % (2 * S)) * 0x9E3779B97F4A7C15ull;
        for (int v = 0; v < N; ++v) {
            uint64_t kv = key ^ ((v0 + v) * 0xD1B54A32D192ED03ull);
            uint32_t u = (uint32_t)(mix64(kv ^ sh) >> 32);
            plane[v] = u < (uint32_t)fmin(floor(p[v] * 4294967296.0), 4294967295.0);
        }
        memset(plane + N, 0, 64);
        int k = 0, ns;
        names[k] = "run only"; tot[k] += greedy(plane, cand_run, 0, &ns); seqs[k++] += ns;
        static HashSt hs;
        for (int kb = 8; kb <= 16; kb += 4) {
            memset(hs.tab, 0xff, sizeof(hs.tab)); hs.keybits = kb;
            static char nm[3][32]; sprintf(nm[(kb - 8) / 4], "hash %d-bit exact + run", kb);
            names[k] = nm[(kb - 8) / 4]; tot[k] += greedy(plane, cand_hash, &hs, &ns); seqs[k++] += ns;
        }
        OffSt os;
        os.maxoff = 64; os.far = 0; names[k] = "offsets 1..64"; tot[k] += greedy(plane, cand_offsets, &os, &ns); seqs[k++] += ns;
        os.maxoff = 128; os.far = 0; names[k] = "offsets 1..128"; tot[k] += greedy(plane, cand_offsets, &os, &ns); seqs[k++] += ns;
        os.maxoff = 64; os.far = 1; names[k] = "offsets 1..64 + k*64"; tot[k] += greedy(plane, cand_offsets, &os, &ns); seqs[k++] += ns;
        os.maxoff = 32; os.far = 0; names[k] = "offsets 1..32"; tot[k] += greedy(plane, cand_offsets, &os, &ns); seqs[k++] += ns;
        os.maxoff = 4095; os.far = 0; names[k] = "all offsets (optimal longest)"; tot[k] += greedy(plane, cand_offsets, &os, &ns); seqs[k++] += ns;
        static EdgeSt es;
        for (int mode = 1; mode <= 2; ++mode)
            for (int kb = 12; kb <= 20; kb += 4) {
                memset(es.tab, 0xff, sizeof(es.tab)); es.keybits = 12; es.mode = mode; es.backext = (kb - 12) * 2;
                static char nm3[6][40]; int j = (mode - 1) * 3 + (kb - 12) / 4;
                sprintf(nm3[j], "edge mode %d, 12-bit, back %d", mode, es.backext);
                names[k] = nm3[j]; tot[k] += greedy(plane, cand_edge, &es, &ns); seqs[k++] += ns;
            }
        static ChainSt cs;
        for (int d = 2; d <= 16; d *= 2) {
            memset(cs.head, 0xff, sizeof(cs.head)); cs.keybits = 12; cs.depth = d;
            static char nm2[4][32]; int j = d == 2 ? 0 : d == 4 ? 1 : d == 8 ? 2 : 3; sprintf(nm2[j], "chain 12-bit depth %d", d);
            names[k] = nm2[j]; tot[k] += greedy(plane, cand_chain, &cs, &ns); seqs[k++] += ns;
        }
    }
    for (int k = 0; k < 32 && names[k]; ++k)
        printf("%-32s ratio %.3f   %.1f sequences / plane  (%.1f B)\n", names[k], (double)planes * N / tot[k], (double)seqs[k] / planes, (double)tot[k] / planes);
    return 0;
}
