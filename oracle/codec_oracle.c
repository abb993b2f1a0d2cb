/*
 * codec_oracle.c — CPU ORACLE (test infrastructure only; see hhgt_oracle.h).
 *
 * Restates the codec the reference reaches through ONE call:
 *   create_dataset(..., compression=32001, compression_opts=(2,2,0,0,5,1,2), chunks=True)
 *   /root/reference/src/haplohyped/vcf_to_h5.py:134-135   (same opts: tests/test_compression.py:45-46,86)
 * Filter id 32001 is the registered id of the Blosc (v1) HDF5 filter — hdf5-blosc / hdf5plugin.Blosc; Blosc2's id
 * is 32026 — although the reference's prose calls its codec "Blosc2" (DESIGN.md §4).  Its cd_values select
 * byte-shuffle (cd_values[5]=1), clevel 5 (cd_values[4]), compcode 2 (cd_values[6]; LZ4HC in Blosc numbering, same
 * block format as LZ4), and the stored chunks are Blosc-1 chunks (16-byte header).  hdf5plugin (>=4.0) and b2h5py
 * are third-party, un-vendored and unpinned (requirements.txt:7-8) and absent here, so the published formats are
 * restated — the Blosc-1 chunk, and the Blosc2 extended-header variant BASELINE.json's wording asks for:
 *   - LZ4 block format (lz4_Block_format.md): token | literal-length ext | literals | offset LE16 |
 *     match-length ext; last 5 bytes literal, last match starts >= 12 bytes before the end.
 *   - Blosc byte shuffle: dst[j*nelem + i] = src[i*typesize + j], tail (nbytes % typesize) verbatim.
 *   - Blosc chunk: header | bstarts[nblocks] int32 | per block, per stream: int32 csize + bytes
 *     (csize == stream size  => stored).  Split into `typesize` streams when the "don't split"
 *     flag (0x10) is clear and the block is not the leftover block.
 * Pinned in tests against liblz4 1.9.3 and c-blosc 1.21 (both present in the image); the 16
 * extra bytes of the Blosc2 extended header are "parity unpinned" (no c-blosc2 in the image).
 */
#include "hhgt_oracle.h"
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ shuffle */
void oracle_shuffle(const uint8_t *src, uint8_t *dst, size_t nbytes, int typesize)
{
    if (typesize <= 1) {
        memcpy(dst, src, nbytes);
        return;
    }
    size_t nelem = nbytes / (size_t)typesize;
    size_t body = nelem * (size_t)typesize;
    for (size_t j = 0; j < (size_t)typesize; ++j)
        for (size_t i = 0; i < nelem; ++i) dst[j * nelem + i] = src[i * (size_t)typesize + j];
    memcpy(dst + body, src + body, nbytes - body);
}

void oracle_unshuffle(const uint8_t *src, uint8_t *dst, size_t nbytes, int typesize)
{
    if (typesize <= 1) {
        memcpy(dst, src, nbytes);
        return;
    }
    size_t nelem = nbytes / (size_t)typesize;
    size_t body = nelem * (size_t)typesize;
    for (size_t i = 0; i < nelem; ++i)
        for (size_t j = 0; j < (size_t)typesize; ++j) dst[i * (size_t)typesize + j] = src[j * nelem + i];
    memcpy(dst + body, src + body, nbytes - body);
}

/* ---------------------------------------------------------------------- LZ4 */
int oracle_lz4_bound(int n) { return n + n / 255 + 16; }

#define MINMATCH 4
#define MFLIMIT 12
#define LASTLITERALS 5
#define HASHLOG 12

static uint32_t rd32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

static uint8_t *put_len(uint8_t *op, int len)
{
    while (len >= 255) {
        *op++ = 255;
        len -= 255;
    }
    *op++ = (uint8_t)len;
    return op;
}

int oracle_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap)
{
    uint8_t *op = dst;
    uint8_t *const oend = dst + cap;
    int anchor = 0;
    if (n >= MFLIMIT + 1) {
        int32_t table[1 << HASHLOG];
        for (int i = 0; i < (1 << HASHLOG); ++i) table[i] = -1;
        const int mflimit = n - MFLIMIT;
        const int matchlimit = n - LASTLITERALS;
        int ip = 0;
        table[(rd32(src) * 2654435761u) >> (32 - HASHLOG)] = 0;
        ip = 1;
        int misses = 0;
        while (ip <= mflimit) {
            uint32_t seq = rd32(src + ip);
            uint32_t h = (seq * 2654435761u) >> (32 - HASHLOG);
            int ref = table[h];
            table[h] = ip;
            if (ref < 0 || ip - ref > 65535 || rd32(src + ref) != seq) {
                ip += 1 + (misses++ >> 6);
                continue;
            }
            misses = 0;
            /* extend backwards */
            while (ip > anchor && ref > 0 && src[ip - 1] == src[ref - 1]) {
                ip--;
                ref--;
            }
            int ml = MINMATCH;
            while (ip + ml < matchlimit && src[ip + ml] == src[ref + ml]) ml++;
            int ll = ip - anchor;
            /* worst-case size of this sequence */
            if (op + 1 + ll / 255 + 1 + ll + 2 + (ml - MINMATCH) / 255 + 1 > oend) return 0;
            uint8_t *token = op++;
            if (ll >= 15) {
                *token = 15 << 4;
                op = put_len(op, ll - 15);
            } else
                *token = (uint8_t)(ll << 4);
            memcpy(op, src + anchor, (size_t)ll);
            op += ll;
            *op++ = (uint8_t)((ip - ref) & 0xff);
            *op++ = (uint8_t)((ip - ref) >> 8);
            if (ml - MINMATCH >= 15) {
                *token |= 15;
                op = put_len(op, ml - MINMATCH - 15);
            } else
                *token |= (uint8_t)(ml - MINMATCH);
            ip += ml;
            anchor = ip;
            if (ip <= mflimit && ip >= 2) table[(rd32(src + ip - 2) * 2654435761u) >> (32 - HASHLOG)] = ip - 2;
        }
    }
    int ll = n - anchor;
    if (op + 1 + ll / 255 + 1 + ll > oend) return 0;
    uint8_t *token = op++;
    if (ll >= 15) {
        *token = 15 << 4;
        op = put_len(op, ll - 15);
    } else
        *token = (uint8_t)(ll << 4);
    memcpy(op, src + anchor, (size_t)ll);
    op += ll;
    return (int)(op - dst);
}

int oracle_lz4_decompress(const uint8_t *src, int csize, uint8_t *dst, int cap)
{
    const uint8_t *ip = src, *iend = src + csize;
    uint8_t *op = dst, *oend = dst + cap;
    if (csize <= 0) return -1;
    for (;;) {
        if (ip >= iend) return -2;
        unsigned token = *ip++;
        size_t ll = token >> 4;
        if (ll == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -3;
                b = *ip++;
                ll += b;
            } while (b == 255);
        }
        if ((size_t)(iend - ip) < ll || (size_t)(oend - op) < ll) return -4;
        memcpy(op, ip, ll);
        op += ll;
        ip += ll;
        if (ip == iend) break; /* last sequence: literals only */
        if (iend - ip < 2) return -5;
        size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        if (off == 0 || off > (size_t)(op - dst)) return -6;
        size_t ml = token & 15;
        if (ml == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -7;
                b = *ip++;
                ml += b;
            } while (b == 255);
        }
        ml += MINMATCH;
        if ((size_t)(oend - op) < ml) return -8;
        const uint8_t *m = op - off;
        for (size_t i = 0; i < ml; ++i) op[i] = m[i]; /* overlapping copy semantics */
        op += ml;
    }
    return (int)(op - dst);
}

/* -------------------------------------------------------------------- Blosc */
#define BLOSC_DOSHUFFLE 0x1
#define BLOSC_MEMCPYED 0x2
#define BLOSC_DOBITSHUFFLE 0x4
#define BLOSC_DONT_SPLIT 0x10
#define BLOSC_LZ4_FORMAT 1
#define BLOSC_MIN_BUFFERSIZE 128
#define BLOSC_MAX_SPLITS 16

static void wr32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}
static uint32_t ld32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static int header_len(int format) { return format == ORACLE_BLOSC2 ? 32 : 16; }

/* c-blosc never writes a blocksize larger than the chunk (its decoder rejects such headers):
 * blocksize = min(blocksize, nbytes), rounded down to a multiple of typesize. */
static int effective_blocksize(size_t nbytes, int typesize, int blocksize)
{
    if ((size_t)blocksize > nbytes) {
        blocksize = (int)nbytes;
        if (typesize > 1 && blocksize >= typesize) blocksize -= blocksize % typesize;
    }
    return blocksize > 0 ? blocksize : 1;
}

static int do_split(int typesize, int blocksize)
{
    return typesize >= 2 && typesize <= BLOSC_MAX_SPLITS && blocksize / typesize >= BLOSC_MIN_BUFFERSIZE;
}

size_t oracle_blosc_bound(size_t nbytes, int typesize, int blocksize)
{
    blocksize = effective_blocksize(nbytes, typesize, blocksize);
    size_t nblocks = (nbytes + (size_t)blocksize - 1) / (size_t)blocksize;
    size_t nstreams = do_split(typesize, blocksize) ? (size_t)typesize : 1;
    return 32 + nblocks * 4 + nblocks * nstreams * 4 + nbytes + 64;
}

static void write_header(uint8_t *dst, int format, int flags, int typesize, size_t nbytes,
                         int blocksize, size_t cbytes)
{
    if (format == ORACLE_BLOSC2) {
        memset(dst, 0, 32);
        dst[0] = 5;                                               /* BLOSC2_VERSION_FORMAT_STABLE */
        dst[1] = 1;                                               /* LZ4 format version           */
        dst[2] = (uint8_t)(flags | BLOSC_DOSHUFFLE | BLOSC_DOBITSHUFFLE); /* both => extended hdr */
        dst[3] = (uint8_t)typesize;
        dst[16 + 5] = (flags & BLOSC_DOSHUFFLE) ? 1 : 0;          /* filters[5] = BLOSC_SHUFFLE   */
    } else {
        dst[0] = 2; /* BLOSC_VERSION_FORMAT */
        dst[1] = 1;
        dst[2] = (uint8_t)flags;
        dst[3] = (uint8_t)typesize;
    }
    wr32(dst + 4, (uint32_t)nbytes);
    wr32(dst + 8, (uint32_t)blocksize);
    wr32(dst + 12, (uint32_t)cbytes);
}

int64_t oracle_blosc_compress(const uint8_t *src, size_t nbytes, int typesize, int blocksize,
                              int format, uint8_t *dst, size_t cap)
{
    if (typesize < 1 || typesize > 255 || blocksize <= 0 || nbytes > 0x7fffffffu) return -1;
    if (typesize > 1 && blocksize % typesize) return -1;
    blocksize = effective_blocksize(nbytes, typesize, blocksize);
    const int hl = header_len(format);
    const int split = do_split(typesize, blocksize);
    int flags = (BLOSC_LZ4_FORMAT << 5) | (typesize > 1 ? BLOSC_DOSHUFFLE : 0) | (split ? 0 : BLOSC_DONT_SPLIT);
    size_t nblocks = (nbytes + (size_t)blocksize - 1) / (size_t)blocksize;
    if (cap < oracle_blosc_bound(nbytes, typesize, blocksize)) return -2;
    uint8_t *tmp = (uint8_t *)malloc((size_t)blocksize);
    uint8_t *cbuf = (uint8_t *)malloc((size_t)oracle_lz4_bound(blocksize));
    size_t op = (size_t)hl + nblocks * 4;
    for (size_t b = 0; b < nblocks; ++b) {
        size_t off = b * (size_t)blocksize;
        size_t bsize = nbytes - off < (size_t)blocksize ? nbytes - off : (size_t)blocksize;
        int leftover = bsize != (size_t)blocksize;
        wr32(dst + hl + 4 * b, (uint32_t)op);
        oracle_shuffle(src + off, tmp, bsize, typesize);
        int nstreams = (split && !leftover) ? typesize : 1;
        size_t neblock = bsize / (size_t)nstreams;
        for (int j = 0; j < nstreams; ++j) {
            int c = oracle_lz4_compress(tmp + (size_t)j * neblock, (int)neblock, cbuf, (int)neblock - 1);
            if (c <= 0 || (size_t)c >= neblock) { /* stored */
                wr32(dst + op, (uint32_t)neblock);
                memcpy(dst + op + 4, tmp + (size_t)j * neblock, neblock);
                op += 4 + neblock;
            } else {
                wr32(dst + op, (uint32_t)c);
                memcpy(dst + op + 4, cbuf, (size_t)c);
                op += 4 + (size_t)c;
            }
        }
    }
    free(tmp);
    free(cbuf);
    if (op > nbytes + (size_t)hl) { /* incompressible: memcpyed chunk */
        flags |= BLOSC_MEMCPYED;
        op = (size_t)hl + nbytes;
        memcpy(dst + hl, src, nbytes);
    }
    write_header(dst, format, flags, typesize, nbytes, blocksize, op);
    return (int64_t)op;
}

int oracle_blosc_info(const uint8_t *chunk, size_t avail, uint32_t *nbytes, uint32_t *blocksize,
                      uint32_t *cbytes, int *typesize, int *flags, int *version)
{
    if (avail < 16) return -1;
    if (version) *version = chunk[0];
    if (flags) *flags = chunk[2];
    if (typesize) *typesize = chunk[3];
    if (nbytes) *nbytes = ld32(chunk + 4);
    if (blocksize) *blocksize = ld32(chunk + 8);
    if (cbytes) *cbytes = ld32(chunk + 12);
    return 0;
}

int64_t oracle_blosc_decompress(const uint8_t *chunk, size_t cbytes_avail, uint8_t *dst, size_t cap)
{
    if (cbytes_avail < 16) return -1;
    int version = chunk[0];
    int flags = chunk[2];
    int typesize = chunk[3];
    size_t nbytes = ld32(chunk + 4);
    size_t blocksize = ld32(chunk + 8);
    size_t cbytes = ld32(chunk + 12);
    int extended = (flags & BLOSC_DOSHUFFLE) && (flags & BLOSC_DOBITSHUFFLE);
    int hl = 16;
    int doshuffle = (flags & BLOSC_DOSHUFFLE) != 0;
    if (extended) {
        if (version < 3 || cbytes_avail < 32) return -2;
        hl = 32;
        doshuffle = 0;
        for (int k = 0; k < 6; ++k) {
            if (chunk[16 + k] == 1) doshuffle = 1;
            else if (chunk[16 + k] != 0) return -3; /* other filters unsupported by the oracle */
        }
    } else if (flags & BLOSC_DOBITSHUFFLE)
        return -3;
    if (((flags >> 5) & 7) != BLOSC_LZ4_FORMAT && !(flags & BLOSC_MEMCPYED)) return -4;
    if (cbytes > cbytes_avail || nbytes > cap) return -5;
    if (flags & BLOSC_MEMCPYED) {
        if (cbytes != nbytes + (size_t)hl) return -6;
        memcpy(dst, chunk + hl, nbytes);
        return (int64_t)nbytes;
    }
    if (blocksize == 0) return nbytes == 0 ? 0 : -7;
    size_t nblocks = (nbytes + blocksize - 1) / blocksize;
    int dont_split = (flags & BLOSC_DONT_SPLIT) != 0;
    uint8_t *tmp = (uint8_t *)malloc(blocksize);
    int64_t rc = (int64_t)nbytes;
    for (size_t b = 0; b < nblocks && rc >= 0; ++b) {
        size_t off = b * blocksize;
        size_t bsize = nbytes - off < blocksize ? nbytes - off : blocksize;
        int leftover = bsize != blocksize;
        size_t sp = ld32(chunk + hl + 4 * b);
        int nstreams = (!dont_split && !leftover) ? typesize : 1;
        size_t neblock = bsize / (size_t)nstreams;
        for (int j = 0; j < nstreams; ++j) {
            if (sp + 4 > cbytes) {
                rc = -8;
                break;
            }
            int32_t cs = (int32_t)ld32(chunk + sp);
            sp += 4;
            if (cs <= 0 || sp + (size_t)cs > cbytes) {
                rc = -9;
                break;
            }
            if ((size_t)cs == neblock)
                memcpy(tmp + (size_t)j * neblock, chunk + sp, neblock);
            else {
                int d = oracle_lz4_decompress(chunk + sp, cs, tmp + (size_t)j * neblock, (int)neblock);
                if (d != (int)neblock) {
                    rc = -10;
                    break;
                }
            }
            sp += (size_t)cs;
        }
        if (rc < 0) break;
        if (doshuffle && typesize > 1)
            oracle_unshuffle(tmp, dst + off, bsize, typesize);
        else
            memcpy(dst + off, tmp, bsize);
    }
    free(tmp);
    return rc;
}
