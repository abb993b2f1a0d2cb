// CPU-only harness for csrc/fast_inflate.h (built with -fsanitize=address,undefined by tests/test_fast_inflate.py):
// deflate streams of every flavour zlib can produce must inflate to the same bytes; corrupted and truncated streams must
// come back as errors or — where zlib itself accepts the bytes — as the same output, and nothing may be touched outside
// the two buffers (the output is surrounded by guard pages' worth of canaries, the input is an exact-size heap block).
#include "../haplohyped_varawareml_amd/csrc/fast_inflate.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <zlib.h>

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

static std::vector<uint8_t> make_text(int kind, size_t n)
{
    std::vector<uint8_t> t(n);
    switch (kind) {
    case 0:   // genotype columns
        for (size_t i = 0; i < n; ++i) t[i] = (i & 3) == 1 ? '|' : (i & 3) == 3 ? '\t' : (rnd() % 16 == 0 ? '1' : '0');
        break;
    case 1:   // random bytes
        for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)rnd();
        break;
    case 2:   // long runs
        for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)('a' + (i / 700) % 3);
        break;
    case 3:   // short periods 1..9
        for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)('0' + i % (1 + (i / 997) % 9));
        break;
    default:  // text with a skewed alphabet
        for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)("ACGTN\t\n0123|./:"[rnd() % (rnd() % 15 + 1)]);
    }
    return t;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &t, int level, int strategy)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, t.size()) + 64);
    zs.next_in = const_cast<Bytef *>(t.data());
    zs.avail_in = (uInt)t.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static int zlib_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    zs.next_in = const_cast<Bytef *>(in);
    zs.avail_in = (uInt)n;
    zs.next_out = out;
    zs.avail_out = (uInt)cap;
    const int rc = inflate(&zs, Z_FINISH);
    *produced = zs.total_out;
    inflateEnd(&zs);
    return rc;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    long ok = 0, bad_ok = 0, total = 0, agree_err = 0;
    for (int r = 0; r < rounds; ++r) {
        const int kind = r % 5;
        const size_t n = r % 7 == 0 ? (size_t)(rnd() % 300) : (size_t)(rnd() % 65536) + 1;
        const std::vector<uint8_t> text = make_text(kind, r % 11 == 0 ? 65280 : n);
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        const int level = r % 10, strategy = strategies[(r / 3) % 5];
        const std::vector<uint8_t> comp = deflate_raw(text, level, strategy);
        // exact-size heap blocks: ASan sees any access past them
        uint8_t *in = (uint8_t *)malloc(comp.size() ? comp.size() : 1);
        memcpy(in, comp.data(), comp.size());
        uint8_t *out = (uint8_t *)malloc(text.size() ? text.size() : 1);
        ++total;
        const int rc = hhgt_fast_inflate_impl(in, comp.size(), out, text.size());
        if (rc != 0 || memcmp(out, text.data(), text.size()) != 0) {
            printf("FAIL: valid stream rejected or wrong (round %d kind %d level %d strategy %d n %zu rc %d)\n", r, kind, level, strategy, text.size(), rc);
            return 1;
        }
        ++ok;
        // wrong expected sizes
        if (text.size() > 1 && hhgt_fast_inflate_impl(in, comp.size(), out, text.size() - 1) == 0) {
            printf("FAIL: accepted a stream longer than the buffer\n");
            return 1;
        }
        free(out);
        // corruptions: bit flips, truncation, random tails
        for (int c = 0; c < 12 && comp.size() > 2; ++c) {
            std::vector<uint8_t> bad = comp;
            if (c % 3 == 0) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (c % 3 == 1) bad.resize(rnd() % bad.size());
            else for (size_t i = bad.size() / 2 + rnd() % (bad.size() / 2 + 1); i < bad.size(); ++i) bad[i] = (uint8_t)rnd();
            uint8_t *bi = (uint8_t *)malloc(bad.size() ? bad.size() : 1);
            memcpy(bi, bad.data(), bad.size());
            uint8_t *bo = (uint8_t *)malloc(text.size() ? text.size() : 1);
            std::vector<uint8_t> zo(text.size() + 1);
            size_t zn = 0;
            const int zrc = zlib_inflate(bi, bad.size(), zo.data(), zo.size(), &zn);
            const int frc = hhgt_fast_inflate_impl(bi, bad.size(), bo, text.size());
            ++total;
            if (frc == 0) {
                // accepted: zlib must accept it too, with the same bytes
                if (zrc != Z_STREAM_END || zn != text.size() || memcmp(bo, zo.data(), text.size()) != 0) {
                    printf("FAIL: accepted a stream zlib rejects or decodes differently (round %d corruption %d)\n", r, c);
                    return 1;
                }
                ++bad_ok;
            } else {
                ++agree_err;
            }
            free(bi);
            free(bo);
        }
        free(in);
    }
    printf("ok: %ld valid streams, %ld corrupted (%ld rejected, %ld harmless and accepted like zlib)\n", ok, total - ok, agree_err, bad_ok);
    return 0;
}
