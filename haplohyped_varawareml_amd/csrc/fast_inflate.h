// fast_inflate.h — raw DEFLATE (RFC 1951) decoder for BGZF members, host side (C++, header only so that the CPU test
// suite can compile it alone, with sanitizers, and fuzz it against zlib).
//
// Why: the north star keeps BGZF inflate on the host cores, and a GPU box of this pool grants 16 CPUs' worth of time.
// zlib 1.2.11 inflates VCF genotype text at ~2 GB/s per core (its match copy moves 1-3 bytes per step whenever the
// distance is shorter than the length — and "0|0\t0|0\t..." is all distance-4 matches), i.e. ~26 GB/s for the box:
// half of what the PCIe link takes.  This decoder is built for that text: one table lookup per symbol from a 64-bit
// bit buffer refilled eight bytes at a time, and match copies that replicate short periods with 32-byte stores.
// htslib (what the reference reads BGZF with, /root/reference/cpp/vcfpp.h:1381,1468) links libdeflate for the same
// reason when it is available.
//
// Contract: hhgt_fast_inflate(in, in_len, out, out_len) returns 0 iff the stream is a complete DEFLATE stream that
// inflates to EXACTLY out_len bytes (BGZF states the size in the member's trailer); any other outcome is < 0.  It
// never reads outside [in, in + in_len) or writes outside [out, out + out_len) — neighbouring members are being
// written by other threads.  On an error the reader falls back to zlib for the member, so zlib stays the arbiter of
// what is corrupt.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hhgt_inflate {

enum { LL_BITS = 11, D_BITS = 8, MAX_BITS = 15, LL_SYMS = 288, D_SYMS = 32 };
// table entry: bits 0-7 code length (bits to drop), 8-15 extra bits, 16-30 value, bit 31 literal;
// special values in bits 8-15: 0xFE end of block, 0xFD sub-table link (value = base, bits 0-7 = primary bits to drop,
// extra low nibble... see build), 0xFF invalid
static const uint32_t E_LIT = 0x80000000u;
enum { X_EOB = 0xFE, X_SUB = 0xFD, X_BAD = 0xFF };

struct Tables {
    uint32_t ll[(1 << LL_BITS) + 1024];   // primary + sub-tables (worst case well below the slack)
    uint32_t d[(1 << D_BITS) + 512];
    uint32_t ll_sub_bits, d_sub_bits;
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t rev_bits(uint32_t c, int n)   // n <= 15
{
    uint32_t x = c;
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0F0Fu) << 4) | ((x >> 4) & 0x0F0Fu);
    x = ((x & 0x00FFu) << 8) | ((x >> 8) & 0x00FFu);
    return x >> (16 - n);
}

// canonical Huffman decode table.  entry_of(sym) gives the entry without its code length.  Returns false for an
// over-subscribed code or (where `need_complete`) an incomplete one; a single-code distance alphabet is accepted
// (RFC 1951 allows it), unused slots are marked invalid.
template <typename F>
static bool build_table(const uint8_t *lens, int nsyms, int tbits, uint32_t *tab, size_t tab_cap, F entry_of, bool need_complete)
{
    int count[MAX_BITS + 1] = {0};
    for (int s = 0; s < nsyms; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1, used = 0;
    for (int l = 1; l <= MAX_BITS; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;   // over-subscribed
        used += count[l];
    }
    if (left > 0 && (need_complete || used != 1)) {
        if (need_complete || used > 1) return false;
    }
    // an incomplete alphabet of ONE code is only what zlib (the arbiter of what is valid here) accepts when that code is
    // one bit long: anything else goes back to zlib, which rejects the stream
    if (left > 0 && used == 1 && count[1] != 1) return false;
    uint32_t next[MAX_BITS + 2];
    uint32_t code = 0;
    for (int l = 1; l <= MAX_BITS; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const size_t primary = (size_t)1 << tbits;
    // symbols in canonical order (by length, then by value)
    uint16_t sorted[LL_SYMS];
    {
        uint32_t offs[MAX_BITS + 2];
        offs[1] = 0;
        for (int l = 1; l <= MAX_BITS; ++l) offs[l + 1] = offs[l] + (uint32_t)count[l];
        for (int s = 0; s < nsyms; ++s)
            if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;
    }
    // primary table, built by doubling: the codes of length l land once in [0, 2^l) at their bit-reversed value, then
    // the filled part is copied behind itself (sequential stores instead of one strided pass per symbol)
    int l0 = 1;
    while (l0 < tbits && !count[l0]) ++l0;
    size_t cur = (size_t)1 << l0;
    for (size_t i = 0; i < cur; ++i) tab[i] = (uint32_t)X_BAD << 8;
    int si = 0;
    for (int l = 1; l < l0; ++l) si += count[l];   // zero
    for (int l = l0; l <= tbits; ++l) {
        uint32_t c = next[l];
        for (int k = 0; k < count[l]; ++k, ++c) {
            const int s = sorted[si++];
            tab[rev_bits(c, l)] = entry_of(s) | (uint32_t)l;
        }
        if (l < tbits) {
            memcpy(tab + cur, tab, cur * sizeof(uint32_t));
            cur <<= 1;
        }
    }
    bool any_long = false;
    for (int l = tbits + 1; l <= MAX_BITS; ++l) any_long = any_long || count[l] != 0;
    if (any_long) {
        // sub-tables: per primary prefix, the longest code under it
        uint8_t sub_len[1 << LL_BITS];
        uint32_t sub_base[1 << LL_BITS];
        memset(sub_len, 0, primary);
        int sj = si;
        for (int l = tbits + 1; l <= MAX_BITS; ++l) {
            uint32_t c = next[l];
            for (int k = 0; k < count[l]; ++k, ++c, ++sj) {
                const uint32_t p = rev_bits(c, l) & (uint32_t)(primary - 1);
                if ((uint8_t)(l - tbits) > sub_len[p]) sub_len[p] = (uint8_t)(l - tbits);
            }
        }
        size_t sub_at = primary;
        for (size_t p = 0; p < primary; ++p) {
            if (!sub_len[p]) continue;
            const size_t n = (size_t)1 << sub_len[p];
            if (sub_at + n > tab_cap) return false;
            sub_base[p] = (uint32_t)sub_at;
            for (size_t i = 0; i < n; ++i) tab[sub_at + i] = (uint32_t)X_BAD << 8;
            tab[p] = ((uint32_t)sub_at << 16) | ((uint32_t)X_SUB << 8) | (uint32_t)(tbits | (sub_len[p] << 4));
            sub_at += n;
        }
        for (int l = tbits + 1; l <= MAX_BITS; ++l) {
            uint32_t c = next[l];
            for (int k = 0; k < count[l]; ++k, ++c) {
                const int s = sorted[si++];
                const uint32_t r = rev_bits(c, l);
                const uint32_t p = r & (uint32_t)(primary - 1);
                const uint32_t hi = r >> tbits;
                const int sl = l - tbits;
                const size_t n = (size_t)1 << sub_len[p];
                const uint32_t e = entry_of(s) | (uint32_t)sl;
                for (size_t i = hi; i < n; i += (size_t)1 << sl) tab[sub_base[p] + i] = e;
            }
        }
    }
    return true;
}

static inline uint32_t ll_entry(int s)
{
    if (s < 256) return E_LIT | ((uint32_t)s << 16);
    if (s == 256) return (uint32_t)X_EOB << 8;
    if (s > 285) return (uint32_t)X_BAD << 8;
    return ((uint32_t)LEN_BASE[s - 257] << 16) | ((uint32_t)LEN_EXTRA[s - 257] << 8);
}
static inline uint32_t d_entry(int s)
{
    if (s > 29) return (uint32_t)X_BAD << 8;
    return ((uint32_t)DIST_BASE[s] << 16) | ((uint32_t)DIST_EXTRA[s] << 8);
}

static inline uint64_t load64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
static inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

// Match copy with at least 64 writable bytes behind out + len.  What costs time in a DEFLATE decoder on this text is
// not the bytes moved but the mispredicted branches per match (length classes, loop trip counts): so the first 48 bytes
// are always written, whatever the length (most matches are shorter), and every distance below 16 goes through ONE
// path — pshufb replicates the period into a 16-byte image, stores advance by the largest multiple of the period.
#if defined(__SSSE3__)
struct PeriodMasks {
    uint8_t m[16][16];
    uint8_t step[16];
    PeriodMasks()
    {
        for (int d = 1; d < 16; ++d) {
            for (int i = 0; i < 16; ++i) m[d][i] = (uint8_t)(i % d);
            step[d] = (uint8_t)((16 / d) * d);
        }
        for (int i = 0; i < 16; ++i) m[0][i] = 0;
        step[0] = 16;
    }
};
static const PeriodMasks g_period_masks;
#endif

static inline __attribute__((always_inline)) void copy_match_fast(uint8_t *dst, uint32_t dist, uint32_t len)
{
    const uint8_t *src = dst - dist;
    uint8_t *const end = dst + len;
#if defined(__x86_64__)
    if (dist >= 16) {
        _mm_storeu_si128((__m128i *)dst, _mm_loadu_si128((const __m128i *)src));
        _mm_storeu_si128((__m128i *)(dst + 16), _mm_loadu_si128((const __m128i *)(src + 16)));
        _mm_storeu_si128((__m128i *)(dst + 32), _mm_loadu_si128((const __m128i *)(src + 32)));
        if (len > 48) {
            dst += 48;
            src += 48;
            do {
                _mm_storeu_si128((__m128i *)dst, _mm_loadu_si128((const __m128i *)src));
                dst += 16;
                src += 16;
            } while (dst < end);
        }
        return;
    }
#if defined(__SSSE3__)
    {
        const __m128i v = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)src),
                                           _mm_loadu_si128((const __m128i *)g_period_masks.m[dist]));
        const uint32_t step = g_period_masks.step[dist];
        _mm_storeu_si128((__m128i *)dst, v);
        _mm_storeu_si128((__m128i *)(dst + step), v);
        _mm_storeu_si128((__m128i *)(dst + 2 * step), v);
        if (len > 3 * step) {   // 2 * step + 16 >= 3 * step bytes are written already
            dst += 3 * step;
            while (dst < end) {
                _mm_storeu_si128((__m128i *)dst, v);
                dst += step;
            }
        }
        return;
    }
#endif
#endif
    if (dist >= 8) {
        do {
            store64(dst, load64(src));
            dst += 8;
            src += 8;
        } while (dst < end);
        return;
    }
    if (dist == 1) {
        memset(dst, src[0], len);
        return;
    }
    // short periods: copy a word, advance by the period (what lies behind the period in each word is overwritten by
    // the following steps)
    do {
        store64(dst, load64(src));
        dst += dist;
        src += dist;
    } while (dst < end);
}

struct Bits {
    const uint8_t *in, *in_end;
    uint64_t buf;
    uint32_t cnt;
};

static inline void refill_fast(Bits &b)   // needs in + 8 <= in_end
{
    b.buf |= load64(b.in) << b.cnt;
    b.in += (63 - b.cnt) >> 3;
    b.cnt |= 56;
}
static inline void refill_safe(Bits &b)
{
    while (b.cnt <= 56 && b.in < b.in_end) {
        b.buf |= (uint64_t)*b.in++ << b.cnt;
        b.cnt += 8;
    }
}
static inline void refill(Bits &b)
{
    if (b.in_end - b.in >= 8) refill_fast(b);
    else refill_safe(b);
}

static int inflate_impl(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    Bits b = {in, in + in_len, 0, 0};
    uint8_t *op = out, *const oend = out + out_len;
    Tables T;
    static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const bool fastcopy = true;
    for (;;) {
        refill(b);
        if (b.cnt < 3) return -1;
        const uint32_t final_ = (uint32_t)b.buf & 1u, type = ((uint32_t)b.buf >> 1) & 3u;
        b.buf >>= 3;
        b.cnt -= 3;
        if (type == 0) {
            // stored: to the byte boundary, LEN / NLEN, bytes
            const uint32_t drop = b.cnt & 7u;
            b.buf >>= drop;
            b.cnt -= drop;
            refill(b);
            if (b.cnt < 32) return -1;
            const uint32_t len = (uint32_t)b.buf & 0xFFFFu, nlen = ((uint32_t)(b.buf >> 16)) & 0xFFFFu;
            if ((len ^ nlen) != 0xFFFFu) return -2;
            b.buf >>= 32;
            b.cnt -= 32;
            // give whole bytes still in the bit buffer back to the input
            b.in -= b.cnt >> 3;
            b.buf = 0;
            b.cnt = 0;
            if ((size_t)(b.in_end - b.in) < len || (size_t)(oend - op) < len) return -3;
            memcpy(op, b.in, len);
            op += len;
            b.in += len;
        } else if (type == 3) {
            return -4;
        } else {
            uint8_t lens[LL_SYMS + D_SYMS];
            int nll, nd;
            if (type == 1) {
                nll = 288;
                nd = 32;
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
            } else {
                refill(b);
                if (b.cnt < 14) return -5;
                nll = 257 + ((int)b.buf & 31);
                nd = 1 + ((int)(b.buf >> 5) & 31);
                const int ncl = 4 + ((int)(b.buf >> 10) & 15);
                b.buf >>= 14;
                b.cnt -= 14;
                if (nll > 286 || nd > 30) return -6;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncl; ++i) {
                    if (b.cnt < 3) {
                        refill(b);
                        if (b.cnt < 3) return -7;
                    }
                    cl[CL_ORDER[i]] = (uint8_t)(b.buf & 7u);
                    b.buf >>= 3;
                    b.cnt -= 3;
                }
                uint32_t cltab[128 + 8];
                if (!build_table(cl, 19, 7, cltab, 128 + 8, [](int s) { return (uint32_t)s << 16; }, true)) {
                    // zlib accepts an incomplete code-length code only in the one-symbol case; keep it simple: defer to zlib
                    return -8;
                }
                int n = 0;
                while (n < nll + nd) {
                    refill(b);
                    const uint32_t e = cltab[b.buf & 127u];
                    const uint32_t l = e & 0xFFu;
                    if (((e >> 8) & 0xFFu) == X_BAD || l == 0 || b.cnt < l + 7) return -9;
                    b.buf >>= l;
                    b.cnt -= l;
                    const int sym = (int)(e >> 16);
                    if (sym < 16) {
                        lens[n++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (n == 0) return -10;
                            val = lens[n - 1];
                            rep = 3 + (int)(b.buf & 3u);
                            b.buf >>= 2;
                            b.cnt -= 2;
                        } else if (sym == 17) {
                            rep = 3 + (int)(b.buf & 7u);
                            b.buf >>= 3;
                            b.cnt -= 3;
                        } else {
                            rep = 11 + (int)(b.buf & 127u);
                            b.buf >>= 7;
                            b.cnt -= 7;
                        }
                        if (n + rep > nll + nd) return -11;
                        while (rep--) lens[n++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return -12;   // no end-of-block code
                // move the distance lengths behind a full 288-symbol literal/length alphabet
                memmove(lens + LL_SYMS, lens + nll, (size_t)nd);
                memset(lens + nll, 0, (size_t)(LL_SYMS - nll));
                memset(lens + LL_SYMS + nd, 0, (size_t)(D_SYMS - nd));
            }
            if (!build_table(lens, LL_SYMS, LL_BITS, T.ll, sizeof(T.ll) / 4, ll_entry, true)) return -13;
            if (!build_table(lens + LL_SYMS, D_SYMS, D_BITS, T.d, sizeof(T.d) / 4, d_entry, false)) return -14;
            // ---- symbols, fast loop: while 16 input bytes and 320 output bytes are left nothing needs a bounds check
            // (a symbol takes at most 48 bits, a match writes at most 258 + 15 bytes); invalid codes fall through to
            // the careful loop below, which reports them
            while (b.in_end - b.in >= 16 && oend - op >= 352) {
                refill_fast(b);
                uint32_t e = T.ll[b.buf & ((1u << LL_BITS) - 1)];
                if (e & E_LIT) {
                    // up to three literals per refill (3 x 15 bits)
                    b.buf >>= e & 0xFFu;
                    b.cnt -= e & 0xFFu;
                    *op++ = (uint8_t)(e >> 16);
                    e = T.ll[b.buf & ((1u << LL_BITS) - 1)];
                    if (e & E_LIT) {
                        b.buf >>= e & 0xFFu;
                        b.cnt -= e & 0xFFu;
                        *op++ = (uint8_t)(e >> 16);
                        e = T.ll[b.buf & ((1u << LL_BITS) - 1)];
                        if (e & E_LIT) {
                            b.buf >>= e & 0xFFu;
                            b.cnt -= e & 0xFFu;
                            *op++ = (uint8_t)(e >> 16);
                            continue;
                        }
                    }
                    refill_fast(b);
                }
                uint32_t x = (e >> 8) & 0xFFu;
                if (x >= X_SUB) break;               // sub-table, end of block or invalid: the careful loop decides
                const uint32_t l = e & 0xFFu;
                const uint32_t len = (e >> 16) + ((uint32_t)(b.buf >> l) & ((1u << x) - 1));
                b.buf >>= l + x;
                b.cnt -= l + x;
                const uint32_t de = T.d[b.buf & ((1u << D_BITS) - 1)];
                const uint32_t dx = (de >> 8) & 0xFFu;
                if (dx >= X_SUB) {
                    // long distance code: undo the length symbol (its bits are gone from the buffer: re-read them)
                    // — simplest is to finish this match here with the careful sub-table walk
                    const uint32_t pb = de & 0xFu, sb = (de >> 4) & 0xFu;
                    if (dx != X_SUB) return -25;
                    const uint32_t de2 = T.d[(de >> 16) + ((b.buf >> pb) & ((1u << sb) - 1))];
                    const uint32_t dl2 = de2 & 0xFFu, dx2 = (de2 >> 8) & 0xFFu;
                    if (dx2 >= X_SUB) return -26;
                    const uint32_t dist2 = (de2 >> 16) + ((uint32_t)(b.buf >> (pb + dl2)) & ((1u << dx2) - 1));
                    b.buf >>= pb + dl2 + dx2;
                    b.cnt -= pb + dl2 + dx2;
                    if (dist2 > (size_t)(op - out)) return -22;
                    copy_match_fast(op, dist2, len);
                    op += len;
                    continue;
                }
                const uint32_t dl = de & 0xFFu;
                const uint32_t dist = (de >> 16) + ((uint32_t)(b.buf >> dl) & ((1u << dx) - 1));
                b.buf >>= dl + dx;
                b.cnt -= dl + dx;
                if (dist > (size_t)(op - out)) return -22;
                copy_match_fast(op, dist, len);
                op += len;
            }
            // ---- symbols, careful loop: the last bytes of the member, sub-tables, end of block, errors
            for (;;) {
                refill(b);
                uint32_t e = T.ll[b.buf & ((1u << LL_BITS) - 1)];
                if (((e >> 8) & 0xFFu) == X_SUB) {
                    const uint32_t pb = e & 0xFu, sb = (e >> 4) & 0xFu;
                    e = T.ll[(e >> 16) + ((b.buf >> pb) & ((1u << sb) - 1))];
                    if (b.cnt < pb) return -15;
                    b.buf >>= pb;
                    b.cnt -= pb;
                }
                uint32_t l = e & 0xFFu;
                if (b.cnt < l) return -16;
                if (e & E_LIT) {
                    if (op >= oend) return -17;
                    b.buf >>= l;
                    b.cnt -= l;
                    *op++ = (uint8_t)(e >> 16);
                    // a second literal without another refill (at least 56 - 15 bits were there)
                    e = T.ll[b.buf & ((1u << LL_BITS) - 1)];
                    l = e & 0xFFu;
                    if ((e & E_LIT) && op < oend && b.cnt >= l && ((e >> 8) & 0xFFu) != X_SUB) {
                        b.buf >>= l;
                        b.cnt -= l;
                        *op++ = (uint8_t)(e >> 16);
                    }
                    continue;
                }
                const uint32_t x = (e >> 8) & 0xFFu;
                if (x == X_EOB) {
                    b.buf >>= l;
                    b.cnt -= l;
                    break;
                }
                if (x >= X_SUB) return -18;   // invalid code / nested link
                b.buf >>= l;
                b.cnt -= l;
                if (b.cnt < x) return -19;
                const uint32_t len = (e >> 16) + ((uint32_t)b.buf & ((1u << x) - 1));
                b.buf >>= x;
                b.cnt -= x;
                if (b.cnt < 32) refill(b);
                uint32_t de = T.d[b.buf & ((1u << D_BITS) - 1)];
                if (((de >> 8) & 0xFFu) == X_SUB) {
                    const uint32_t pb = de & 0xFu, sb = (de >> 4) & 0xFu;
                    de = T.d[(de >> 16) + ((b.buf >> pb) & ((1u << sb) - 1))];
                    if (b.cnt < pb) return -20;
                    b.buf >>= pb;
                    b.cnt -= pb;
                }
                const uint32_t dl = de & 0xFFu, dx = (de >> 8) & 0xFFu;
                if (dx >= X_SUB || b.cnt < dl + dx) return -21;
                b.buf >>= dl;
                b.cnt -= dl;
                const uint32_t dist = (de >> 16) + ((uint32_t)b.buf & ((1u << dx) - 1));
                b.buf >>= dx;
                b.cnt -= dx;
                if (dist > (size_t)(op - out)) return -22;
                if ((size_t)(oend - op) < len) return -23;
                if (fastcopy && (size_t)(oend - op) >= (size_t)len + 64) {
                    copy_match_fast(op, dist, len);
                    op += len;
                } else {
                    const uint8_t *s = op - dist;
                    for (uint32_t i = 0; i < len; ++i) op[i] = s[i];
                    op += len;
                }
            }
        }
        if (final_) break;
    }
    if (op != oend) return -24;
    return 0;
}

}  // namespace hhgt_inflate

static inline int hhgt_fast_inflate_impl(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    return hhgt_inflate::inflate_impl(in, in_len, out, out_len);
}
