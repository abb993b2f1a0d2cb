// inflate.hip — BGZF members inflated on gfx950 (outside the north star, which leaves BGZF on the host).
//
// What it replaces: htslib's bgzf_read_block/inflate pair under the reference's reader
// (/root/reference/cpp/vcfpp.h:1381 hts_open, :1468 bcf/vcf read) — RFC 1951 DEFLATE inside RFC 1952 members
// carrying the "BC" extra field.  BGZF members are independent (<= 64 KiB each), so the unit of parallelism is
// the member: one wave per member, wave-uniform control (bit buffer, positions and lengths live in SGPRs),
// lanes used where DEFLATE offers width:
//   * Huffman decode is canonical and table-free: lane L (1..15) tests whether the next L bits are a code of
//     length L (`code - first[L] < count[L]`), one ballot picks the length, one readlane the symbol index;
//   * table construction (counts, first codes, symbols sorted by code) is ballots over 64 code lengths at a time;
//   * match copies move 64 bytes per step (matches of VCF genotype text are long: "0|0\t" x 64);
//   * the compressed input reaches the bit buffer through scalar loads (read-only data, uniform address), one dword
//     ahead, so a refill never waits behind the byte stores of earlier symbols (vector loads share vmcnt with them).
// The output window is the member's slice of the destination itself (deflate distances never leave the member).
// ISIZE and the bit budget of each member are checked by the inflate kernel, the CRC-32 of its text by a second one.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct HuffLane {      // lane L (1..15): canonical-code data of code length L; other lanes: cnt = 0
    uint32_t first, cnt, offs;
};

// lens[0, nsym) (LDS) -> per-length lane data + syms[] (LDS) = symbols sorted by (length, value).  false: over-subscribed.
template <int MAXLEN>
__device__ __forceinline__ bool huff_build(const uint8_t *lens, uint32_t nsym, uint16_t *syms, HuffLane &t, uint32_t lane)
{
    uint32_t tot[MAXLEN + 1];
#pragma unroll
    for (int L = 0; L <= MAXLEN; ++L) tot[L] = 0;
    for (uint32_t base = 0; base < nsym; base += 64u) {
        const uint32_t s = base + lane;
        const uint32_t l = s < nsym ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L <= MAXLEN; ++L) tot[L] += (uint32_t)__builtin_popcountll(__ballot(l == (uint32_t)L));
    }
    uint32_t start[MAXLEN + 1];
    uint32_t code = 0, off = 0;
    bool ok = true;
    t.first = 0;
    t.cnt = 0;
    t.offs = 0;
#pragma unroll
    for (int L = 1; L <= MAXLEN; ++L) {
        code = (code + tot[L - 1]) << 1;  // tot[0] stays 0: unused symbols take no code space
        if (code + tot[L] > (1u << L)) ok = false;
        start[L] = off;
        if (lane == (uint32_t)L) {
            t.first = code;
            t.cnt = tot[L];
            t.offs = off;
        }
        off += tot[L];
    }
    for (uint32_t base = 0; base < nsym; base += 64u) {
        const uint32_t s = base + lane;
        const uint32_t l = s < nsym ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L <= MAXLEN; ++L) {
            const unsigned long long m = __ballot(l == (uint32_t)L);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (l == (uint32_t)L) syms[start[L] + rank] = (uint16_t)s;
            start[L] += (uint32_t)__builtin_popcountll(m);
        }
    }
    return ok;
}

// next symbol of the code in t from the low bits of `bits`; nbits = its length.  -1: no code matches.
// As huff_decode, with the first 64 sorted symbols also held one per lane in `front` (the short, frequent codes
// sort first): a hit there is a second readlane instead of an LDS round trip.
__device__ __forceinline__ int huff_decode_front(const HuffLane &t, const uint16_t *syms, uint32_t front, uint32_t bits,
                                                 uint32_t shiftv, uint32_t &nbits)
{
    const uint32_t rev = __builtin_bitreverse32(bits);
    const uint32_t d = (rev >> shiftv) - t.first;
    const unsigned long long m = __ballot(d < t.cnt);
    if (m == 0ull) return -1;
    const uint32_t L = (uint32_t)__builtin_ctzll(m);
    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)(t.offs + d), (int)L);
    nbits = L;
    if (idx < 64u) return __builtin_amdgcn_readlane((int)front, (int)idx);
    return (int)sgpr(syms[idx]);
}

__device__ __forceinline__ int huff_decode(const HuffLane &t, const uint16_t *syms, uint32_t bits, uint32_t shiftv,
                                           uint32_t &nbits)
{
    const uint32_t rev = __builtin_bitreverse32(bits);  // deflate packs codes MSB-first into an LSB-first stream
    const uint32_t d = (rev >> shiftv) - t.first;        // lane L: the next L bits as a code, relative to the first of that length
    const unsigned long long m = __ballot(d < t.cnt);
    if (m == 0ull) return -1;
    const uint32_t L = (uint32_t)__builtin_ctzll(m);     // a prefix code matches at exactly one length; lowest is it
    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)(t.offs + d), (int)L);
    nbits = L;
    return (int)sgpr(syms[idx]);
}

enum : uint32_t {
    INF_OK = 0,
    INF_BAD_BLOCK_TYPE = 1,
    INF_BAD_STORED = 2,
    INF_BAD_TABLE = 3,
    INF_BAD_CODE = 4,
    INF_BAD_DISTANCE = 5,
    INF_OUTPUT_OVERRUN = 6,
    INF_INPUT_OVERRUN = 7,
    INF_SIZE_MISMATCH = 8,
    INF_CRC_MISMATCH = 9,
};

// order in which the code-length code lengths are stored (RFC 1951 3.2.7), 5 bits each, packed
__device__ __forceinline__ uint32_t clc_order(uint32_t i)
{
    // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) |
                                  (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    return i < 12u ? (uint32_t)(lo >> (5u * i)) & 31u : (uint32_t)(hi >> (5u * (i - 12u))) & 31u;
}

}  // namespace

// grid = ceil(n_members / 4); block = 256 (wave w of a block inflates member 4 * blockIdx + w; no barriers)
#ifndef INF_WGS
#define INF_WGS 7   // workgroups per CU the kernel is compiled for: 72 registers + 36 bytes of scratch per lane, seven waves per SIMD —
                    // 210 against 198 GB/s of text at 18 k members (6: 74 registers, no scratch; 8: slower, round 3)
#endif
__global__ __launch_bounds__(256, INF_WGS) void k_inflate_members(const uint8_t *__restrict__ src, uint64_t src_bytes,
                                                         const unsigned long long *__restrict__ comp_off,
                                                         const uint32_t *__restrict__ comp_len,
                                                         const unsigned long long *__restrict__ out_off,
                                                         const uint32_t *__restrict__ isize, uint32_t n_members,
                                                         uint8_t *dst, uint64_t dst_bytes, uint32_t *__restrict__ status)
{
    __shared__ uint8_t s_lens[4][352];  // 19 code-length-code lengths + up to 316 code lengths while a dynamic header is read
    __shared__ uint16_t s_ll[4][288];
    __shared__ uint16_t s_dd[4][32];
    const uint32_t wave = sgpr(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t mem = blockIdx.x * 4u + wave;
    if (mem >= n_members) return;
    uint8_t *lens = s_lens[wave];
    uint16_t *ll_syms = s_ll[wave];
    uint16_t *dd_syms = s_dd[wave];

    const uint64_t p0 = comp_off[mem];
    const uint32_t clen = comp_len[mem];
    const uint32_t osz = isize[mem];
    uint8_t *out = dst + out_off[mem];
    const uint32_t *srcw = reinterpret_cast<const uint32_t *>(src);
    const uint32_t last_word = (uint32_t)(src_bytes >> 2) - 1u;  // src_bytes is a multiple of 4 (checked by the launcher)
    const uint32_t end_word = (uint32_t)((p0 + clen + 3u) >> 2);  // first dword wholly behind the member's payload
    const uint32_t shiftv = (lane >= 1u && lane <= 15u) ? 32u - lane : 31u;

    // ---- compressed input: one dword ahead of the bit buffer, fetched with scalar loads (the payload is read-only
    // for the kernel, the address is wave-uniform).  Scalar loads count in lgkmcnt: a refill never waits for the
    // byte stores of earlier symbols, which a vector-load window (vmcnt) did at every refill.
    uint32_t wi = (uint32_t)(p0 >> 2);  // dword index into src (the launcher keeps src_bytes below 16 GiB)
    uint32_t ahead = 0;
    auto sload = [&](uint32_t w) -> uint32_t {
        const uint32_t *q = srcw + (w < last_word ? w : last_word);
        uint32_t v;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
        return v;
    };
    auto fetch = [&]() -> uint32_t {  // dword wi, then wi += 1; the dword after it is requested for the next call
        const uint32_t v = ahead;
        ++wi;
        ahead = sload(wi);
        return v;
    };
    uint64_t bb;
    uint32_t bc;
    auto start_at = [&](uint64_t byte_pos) {
        wi = (uint32_t)(byte_pos >> 2);
        ahead = sload(wi);
        const uint32_t mis = (uint32_t)byte_pos & 3u;
        bb = (uint64_t)(fetch() >> (8u * mis));
        bc = 32u - 8u * mis;
    };
    auto refill = [&]() {  // afterwards bc >= 32
        if (bc < 32u) {
            bb |= (uint64_t)fetch() << bc;
            bc += 32u;
        }
    };
    auto take = [&](uint32_t n) -> uint32_t {  // n <= 16, bc >= n
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    };
    uint32_t op = 0, err = INF_OK;
    bool last = false;
    if (p0 + clen > src_bytes || out_off[mem] + osz > dst_bytes) {  // table entries that leave the buffers
        if (lane == 0u) status[mem] = INF_INPUT_OVERRUN;
        return;
    }
    start_at(p0);

    while (!last && err == INF_OK) {
        if (wi > end_word + 2u) {  // reading far behind the payload: malformed (also bounds every loop below)
            err = INF_INPUT_OVERRUN;
            break;
        }
        refill();
        last = take(1) != 0u;
        const uint32_t btype = take(2);
        if (btype == 3u) {
            err = INF_BAD_BLOCK_TYPE;
            break;
        }
        if (btype == 0u) {  // stored: skip to the byte boundary, LEN / ~LEN, raw bytes
            take(bc & 7u);
            refill();
            const uint32_t len = take(16), nlen = take(16);
            const uint64_t bp = ((uint64_t)wi << 2) - (bc >> 3);  // first byte not yet consumed
            if (len != (~nlen & 0xFFFFu) || bp + len > p0 + clen || len > osz - op) {
                err = INF_BAD_STORED;
                break;
            }
            for (uint32_t k = lane; k < len; k += 64u) out[op + k] = src[bp + k];
            op += len;
            start_at(bp + len);
            continue;
        }
        HuffLane LL, DD;
        if (btype == 1u) {  // fixed code (RFC 1951 3.2.6)
            for (uint32_t s = lane; s < 288u; s += 64u) lens[s] = s < 144u ? 8 : (s < 256u ? 9 : (s < 280u ? 7 : 8));
            if (lane < 32u) lens[288u + lane] = lane < 30u ? 5 : 0;
            huff_build<15>(lens, 288u, ll_syms, LL, lane);
            huff_build<15>(lens + 288, 32u, dd_syms, DD, lane);
        } else {  // dynamic code (3.2.7)
            refill();
            const uint32_t hlit = take(5) + 257u, hdist = take(5) + 1u, hclen = take(4) + 4u;
            if (hlit > 286u || hdist > 30u) {
                err = INF_BAD_TABLE;
                break;
            }
            if (lane < 19u) lens[lane] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                refill();
                const uint32_t l = take(3);
                if (lane == 0u) lens[clc_order(i)] = (uint8_t)l;
            }
            HuffLane CL;
            if (!huff_build<7>(lens, 19u, dd_syms, CL, lane)) {
                err = INF_BAD_TABLE;
                break;
            }
            // code lengths of both alphabets, run-length coded with the code-length code; they land behind the 19
            uint8_t *cl = lens + 19;  // [0, hlit + hdist) <= 316
            const uint32_t ntot = hlit + hdist;
            uint32_t i = 0, prev = 0;
            while (i < ntot && err == INF_OK) {
                if (wi > end_word + 2u) {
                    err = INF_INPUT_OVERRUN;
                    break;
                }
                refill();
                uint32_t nb;
                const int sym = huff_decode(CL, dd_syms, (uint32_t)bb, shiftv, nb);
                if (sym < 0) {
                    err = INF_BAD_TABLE;
                    break;
                }
                take(nb);
                if (sym < 16) {
                    if (lane == 0u) cl[i] = (uint8_t)sym;
                    prev = (uint32_t)sym;
                    ++i;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0u) {
                        err = INF_BAD_TABLE;
                        break;
                    }
                    val = prev;
                    rep = 3u + take(2);
                } else if (sym == 17) {
                    rep = 3u + take(3);
                } else {
                    rep = 11u + take(7);
                }
                if (rep > ntot - i) {
                    err = INF_BAD_TABLE;
                    break;
                }
                for (uint32_t k = lane; k < rep; k += 64u) cl[i + k] = (uint8_t)val;
                prev = val;
                i += rep;
            }
            if (err != INF_OK) break;
            // literal/length lengths -> lens[0..), distance lengths behind them at 288 (moved front to back safely:
            // the source [19, 19 + ntot) is read whole into registers first)
            uint32_t keep[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const uint32_t s = (uint32_t)c * 64u + lane;
                keep[c] = s < ntot ? cl[s] : 0u;
            }
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const uint32_t s = (uint32_t)c * 64u + lane;
                if (s < hlit) lens[s] = (uint8_t)keep[c];
                else if (s < ntot) lens[288u + (s - hlit)] = (uint8_t)keep[c];
            }
            if (sgpr(lens[256]) == 0u) {  // no end-of-block code
                err = INF_BAD_TABLE;
                break;
            }
            if (!huff_build<15>(lens, hlit, ll_syms, LL, lane) || !huff_build<15>(lens + 288, hdist, dd_syms, DD, lane)) {
                err = INF_BAD_TABLE;
                break;
            }
        }
        // ---- symbols of this block ----
        const uint32_t ll_front = ll_syms[lane], dd_front = dd_syms[lane & 31u];
        for (;;) {
            if (wi > end_word + 2u) {
                err = INF_INPUT_OVERRUN;
                break;
            }
            refill();
            uint32_t nb;
            const int sym = huff_decode_front(LL, ll_syms, ll_front, (uint32_t)bb, shiftv, nb);
            if (sym < 0) {
                err = INF_BAD_CODE;
                break;
            }
            bb >>= nb;
            bc -= nb;
            if (sym < 256) {
                if (op >= osz) {
                    err = INF_OUTPUT_OVERRUN;
                    break;
                }
                if (lane == 0u) out[op] = (uint8_t)sym;
                ++op;
                continue;
            }
            if (sym == 256) break;
            const uint32_t li = (uint32_t)sym - 257u;
            if (li > 28u) {
                err = INF_BAD_CODE;
                break;
            }
            uint32_t len;
            if (li < 8u) len = 3u + li;
            else if (li == 28u) len = 258u;
            else {
                const uint32_t eb = (li >> 2) - 1u;  // 1..5 extra bits; bc >= 32 - 15 here
                len = ((4u + (li & 3u)) << eb) + 3u + take(eb);
            }
            refill();
            const int ds = huff_decode_front(DD, dd_syms, dd_front, (uint32_t)bb, shiftv, nb);
            if (ds < 0 || ds > 29) {
                err = INF_BAD_CODE;
                break;
            }
            bb >>= nb;
            bc -= nb;
            uint32_t dist;
            if (ds < 4) dist = (uint32_t)ds + 1u;
            else {
                const uint32_t eb = ((uint32_t)ds >> 1) - 1u;  // 1..13 extra bits; bc >= 32 - 15 here
                dist = ((2u + ((uint32_t)ds & 1u)) << eb) + 1u + take(eb);
            }
            if (dist > op) {
                err = INF_BAD_DISTANCE;
                break;
            }
            if (len > osz - op) {
                err = INF_OUTPUT_OVERRUN;
                break;
            }
            // out[op, op + len) = the len bytes starting dist back; 64 bytes per step.  A step never reads a byte it
            // writes: either dist >= 64, or the source index is folded into the dist bytes before op.
            const uint8_t *pat = out + op - dist;
            if (dist >= 64u || len <= dist) {
                for (uint32_t k = lane; k < len; k += 64u) out[op + k] = pat[k];
            } else {
                uint32_t ph, step;
                if ((dist & (dist - 1u)) == 0u) {
                    ph = lane & (dist - 1u);
                    step = 0u;
                } else {
                    ph = lane % dist;
                    step = 64u % dist;
                }
                for (uint32_t k = lane; k < len; k += 64u) {
                    out[op + k] = pat[ph];
                    ph += step;
                    if (ph >= dist) ph -= dist;
                }
            }
            op += len;
        }
    }
    if (err == INF_OK) {
        // bits consumed must fit the payload, and the member must have produced exactly ISIZE bytes
        const uint64_t used_bits = (((uint64_t)wi << 2) - p0) * 8u - bc;
        if (used_bits > (uint64_t)clen * 8u) err = INF_INPUT_OVERRUN;
        else if (op != osz) err = INF_SIZE_MISMATCH;
    }
    if (lane == 0u) status[mem] = err;
}

// ---- CRC-32 of the inflated members (RFC 1952 trailer; htslib's bgzf.c rejects a member whose CRC differs) ----
// One wave per member again, coalesced (see k_crc32_members); the GF(2) arithmetic is zlib's crc32_combine
// (multmodp / x2nmodp, reflected polynomial 0xEDB88320).
namespace {
constexpr uint32_t CRC_POLY = 0xEDB88320u;

__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b)  // a(x) * b(x) mod P, reflected bit order
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
}  // namespace

// x2n[k] = x^(2^k) mod P.  grid = ceil(n_members / 4), block = 256.
// The text is walked in tiles of 4 KiB: the wave loads a tile with coalesced dwords into LDS (17-dword rows, so
// that the 64-byte pieces the lanes then read are bank-conflict free), lane i runs the bytewise table CRC over piece i
// of the tile.  CRC is linear over GF(2): with s_i the raw CRC (init 0) of "the message with every byte outside lane
// i's pieces zeroed", raw(M) = XOR_i s_i.  Between two pieces of a lane lie 4032 zero bytes = one multiplication by
// x^(8 * 4032); behind its last byte lie (n - end_i) zero bytes = one multiplication by x^(8 (n - end_i)).  Finally
// crc32(M) = ~(raw(M) ^ 0xFFFFFFFF * x^(8 n)).
__global__ __launch_bounds__(256) void k_crc32_members(const uint8_t *__restrict__ text,
                                                       const unsigned long long *__restrict__ out_off,
                                                       const uint32_t *__restrict__ isize,
                                                       const uint32_t *__restrict__ want, uint32_t n_members,
                                                       const uint32_t *__restrict__ x2n, uint32_t *status)
{
    // four tables (slicing by 4: a dword per step, its four look-ups independent of each other, where the bytewise form waits
    // for one LDS round trip per byte): 821 against 865 us per 18 k members alone — the kernel is not only that chain
    __shared__ uint32_t s_tab[4][256];
    __shared__ uint32_t s_x2n[32];
    __shared__ uint32_t s_tile[4][64 * 17];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        s_tab[0][threadIdx.x] = c;
        if (threadIdx.x < 32u) s_x2n[threadIdx.x] = x2n[threadIdx.x];
    }
    __syncthreads();
    {
        uint32_t c = s_tab[0][threadIdx.x];
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            c = (c >> 8) ^ s_tab[0][c & 0xFFu];
            s_tab[t][threadIdx.x] = c;
        }
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t mem = blockIdx.x * 4u + wave;
    if (mem >= n_members) return;  // (no barrier below)
    if (status[mem] != 0u) return;  // not inflated: nothing to check
    const uint32_t n = isize[mem];
    const uint8_t *p = text + out_off[mem];
    uint32_t *tile = s_tile[wave];
    auto x_pow_bytes = [&](uint32_t nbytes) {  // x^(8 * nbytes) mod P
        uint32_t f = 0x80000000u;              // x^0
        for (uint32_t k = 0; k < 17u; ++k)
            if ((nbytes >> k) & 1u) f = crc_multmodp(s_x2n[k + 3u], f);
        return f;
    };
    const uint32_t gap = x_pow_bytes(4032u);
    typedef uint32_t u32u __attribute__((aligned(1)));
    uint32_t s = 0, end = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 4096u) {
        const uint32_t tn = n - t0 < 4096u ? n - t0 : 4096u;  // bytes of this tile
        // tile -> LDS: dword g of the tile (g = 64 j + lane) lands in row g / 16, column g % 16
        if (tn == 4096u) {
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                const uint32_t g = 64u * j + lane;
                tile[(g >> 4) * 17u + (g & 15u)] = *reinterpret_cast<const u32u *>(p + t0 + 4u * g);
            }
        } else {
            for (uint32_t j = 0; j < 16u; ++j) {
                const uint32_t g = 64u * j + lane, o = 4u * g;
                uint32_t w = 0;
                if (o + 4u <= tn) w = *reinterpret_cast<const u32u *>(p + t0 + o);
                else
                    for (uint32_t k = 0; o + k < tn; ++k) w |= (uint32_t)p[t0 + o + k] << (8u * k);
                tile[(g >> 4) * 17u + (g & 15u)] = w;
            }
        }
        // piece of this lane: bytes [64 lane, 64 lane + 64) of the tile, cut at tn
        const uint32_t lo = 64u * lane;
        if (lo < tn) {
            const uint32_t m = tn - lo < 64u ? tn - lo : 64u;
            s = crc_multmodp(gap, s);  // the 4032 zero bytes since this lane's previous piece (s = 0 before the first)
            const uint32_t *row = tile + lane * 17u;
            uint32_t k = 0;
            for (; k + 4u <= m; k += 4u) {   // whole dwords
                const uint32_t x = s ^ row[k >> 2];
                s = s_tab[3][x & 0xFFu] ^ s_tab[2][(x >> 8) & 0xFFu] ^ s_tab[1][(x >> 16) & 0xFFu] ^ s_tab[0][x >> 24];
            }
            for (; k < m; ++k) {             // the last 1-3 bytes of a member
                const uint32_t b = (row[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
                s = s_tab[0][(s ^ b) & 0xFFu] ^ (s >> 8);
            }
            end = t0 + lo + m;
        }
    }
    uint32_t v = s ? crc_multmodp(x_pow_bytes(n - end), s) : 0u;
    for (int d = 32; d >= 1; d >>= 1) v ^= (uint32_t)__shfl_xor((int)v, d, 64);
    const uint32_t crc = ~(v ^ crc_multmodp(x_pow_bytes(n), 0xFFFFFFFFu));
    if (lane == 0u && crc != want[mem]) status[mem] = 9u;  // INF_CRC_MISMATCH
}

int launch_inflate(const uint8_t *d_src, uint64_t src_bytes, const uint64_t *d_comp_off, const uint32_t *d_comp_len,
                   const uint64_t *d_out_off, const uint32_t *d_isize, uint64_t n_members, uint8_t *d_dst,
                   uint64_t dst_bytes, uint32_t *d_status, const uint32_t *d_crc32, const uint32_t *d_x2n, hipStream_t st)
{
    if (n_members == 0) return HHGT_OK;
    if (src_bytes < 4 || (src_bytes & 3u) || src_bytes >= (1ull << 34) || (reinterpret_cast<uintptr_t>(d_src) & 3u)) {
        hhgt_set_error("inflate: the compressed buffer must be 4-byte aligned, a multiple of 4 bytes long and below 16 GiB");
        return HHGT_ERR_ARG;
    }
    const uint64_t grid = (n_members + 3) / 4;
    if (grid > 0x7fffffffull || n_members > 0xffffffffull) {
        hhgt_set_error("inflate: too many members");
        return HHGT_ERR_ARG;
    }
    hipLaunchKernelGGL(k_inflate_members, dim3((uint32_t)grid), dim3(256), 0, st, d_src, src_bytes,
                       reinterpret_cast<const unsigned long long *>(d_comp_off), d_comp_len,
                       reinterpret_cast<const unsigned long long *>(d_out_off), d_isize, (uint32_t)n_members, d_dst,
                       dst_bytes, d_status);
    HIP_TRY(hipGetLastError());
    if (d_crc32) {
        hipLaunchKernelGGL(k_crc32_members, dim3((uint32_t)grid), dim3(256), 0, st, d_dst,
                           reinterpret_cast<const unsigned long long *>(d_out_off), d_isize, d_crc32, (uint32_t)n_members,
                           d_x2n, d_status);
        HIP_TRY(hipGetLastError());
    }
    return HHGT_OK;
}

// x^(2^k) mod P for k = 0..31 (host; uploaded once per context)
void crc32_x2n_table(uint32_t *t)
{
    uint32_t p = 0x40000000u;  // x^1
    t[0] = p;
    for (int k = 1; k < 32; ++k) t[k] = p = crc_multmodp(p, p);
}

// ---- host side: member table of a BGZF byte range (RFC 1952 member with the 6-byte "BC" extra subfield) ----
extern "C" int hhgt_bgzf_scan(const void *host, uint64_t nbytes, uint64_t max_members, uint64_t *comp_off,
                              uint32_t *comp_len, uint32_t *isize, uint32_t *crc32, uint64_t *n_members,
                              uint64_t *consumed)
{
    if (!host || !n_members || !consumed || (max_members && (!comp_off || !comp_len || !isize))) return HHGT_ERR_ARG;
    const uint8_t *p = static_cast<const uint8_t *>(host);
    uint64_t pos = 0, n = 0;
    while (n < max_members && nbytes - pos >= 18) {
        const uint8_t *h = p + pos;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
            hhgt_set_error("bgzf scan: no gzip member with an extra field at byte %llu", (unsigned long long)pos);
            return HHGT_ERR_MALFORMED;
        }
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (nbytes - pos < 12ull + xlen) break;  // header itself is cut: needs more input
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *f = h + 12 + x;
            const uint32_t slen = f[2] | (f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) {
                bsize = (f[4] | (f[5] << 8)) + 1u;
                found = true;
                break;
            }
            x += 4 + slen;
        }
        if (!found || bsize < 12u + xlen + 8u) {
            hhgt_set_error("bgzf scan: member at byte %llu carries no valid BC subfield", (unsigned long long)pos);
            return HHGT_ERR_MALFORMED;
        }
        if (nbytes - pos < bsize) break;  // member is cut: needs more input
        comp_off[n] = pos + 12u + xlen;
        comp_len[n] = bsize - (12u + xlen) - 8u;
        const uint8_t *t = h + bsize - 4;
        isize[n] = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        if (crc32) crc32[n] = t[-4] | (t[-3] << 8) | (t[-2] << 16) | ((uint32_t)t[-1] << 24);
        if (isize[n] > 65536u) {
            hhgt_set_error("bgzf scan: member at byte %llu claims %u bytes (BGZF allows 65536)", (unsigned long long)pos, isize[n]);
            return HHGT_ERR_MALFORMED;
        }
        ++n;
        pos += bsize;
    }
    *n_members = n;
    *consumed = pos;
    return HHGT_OK;
}
