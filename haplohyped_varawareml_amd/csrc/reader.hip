// reader.hip — host-side VCF text reader (see include/hhgt_reader.h): mmap + zlib inflate into a ring of pinned,
// line-aligned blocks; hipMemcpyAsync to HBM.  Host C++ only (compiled by hipcc with the rest of libhhgt.so); no
// device code in this file.
//
// BGZF (the format htslib reads under /root/reference/cpp/vcfpp.h:1381,1468): the scanner thread walks the member
// headers — every member's inflated size sits in its trailer, so the place of every member's text in the ring is
// known before anything is inflated — and hands the members to a pool of inflate workers as per-block task lists.
// There is NO barrier per block: workers move from one block's list to the next while the scanner is already laying
// out later blocks, and a block is published to the consumer when its last task finishes.  The one dependency
// between blocks, the partial last line that moves to the front of the next block, is resolved by the scanner
// itself: it inflates the block's LAST member right away (30 us of zlib), finds the last newline in it and thereby
// knows where the next block's members start.  (Round 1 filled one block at a time with a wake-all / wait-all
// around it: 48 threads delivered 20 GB/s of text where one delivers 2.2.)
// Every member's text is hashed and compared with the CRC-32 in its trailer, as htslib's bgzf.c does; the hash is
// the carry-less-multiply folding form (PCLMULQDQ), an order of magnitude faster than zlib 1.2.11's table crc32()
// so the check no longer costs as much as the inflate it guards.  Plain gzip streams through one inflater,
// uncompressed files are pread into the ring by the same worker pool.
#include "common.h"
#include "../../include/hhgt_reader.h"
#include "fast_inflate.h"
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

struct Task {
    const uint8_t *src = nullptr;
    uint32_t src_len = 0;
    uint8_t *dst = nullptr;
    uint32_t dst_len = 0;
    int fd = -1;            // >= 0: an uncompressed piece, read with pread(fd, dst, dst_len, file_off)
    uint64_t file_off = 0;
    bool check_crc = false;  // BGZF member: compare crc32(dst) with the trailer behind src
};

// one ring block: its task list is written by the scanner before the block is opened to the workers
struct Block {
    uint8_t *buf = nullptr;
    bool pinned = false;
    size_t n = 0;                      // published bytes (whole lines)
    std::vector<Task> tasks;
    // generation << 32 | next task to draw.  Workers advance it by compare-and-swap against the generation they were
    // given with the block, so a worker that still holds a recycled block can neither run a task of its next life
    // nor swallow one of its tickets
    std::atomic<uint64_t> ticket{0};
    std::atomic<uint32_t> done{0};     // tasks finished
    uint32_t gen = 0;
    uint32_t n_tasks = 0;
    bool scanned = false;              // layout complete (n, n_tasks final)
    bool last = false;                 // last block of the input
};

}  // namespace

struct hhgt_reader {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    size_t in_pos = 0;  // next unread compressed byte
    bool is_gzip = false, is_bgzf = false;
    bool check_crc = true;   // HHGT_BGZF_NO_CRC=1 skips the per-member CRC32 (htslib always checks it)
    size_t block_bytes = 0;
    bool pinned = false;
    int n_blocks = 0;
    std::unique_ptr<Block[]> ring;
    // block life cycle: free_ -> (scanner) open_ (workers draw tasks) -> order (publication order) -> held by the
    // consumer -> free_
    std::deque<int> free_, open_, order;
    std::vector<char> held;          // per block: 1 while the consumer holds it
    int auto_held = -1;              // block handed out by hhgt_reader_next (released by the following call)
    bool eof = false, stop = false;  // eof: the scanner has laid out the last block
    int error = 0;
    std::string errmsg;
    std::mutex mu;                   // guards the deques, eof/stop/error
    std::condition_variable cv_free, cv_open, cv_pub;
    std::thread producer;
    std::vector<std::thread> workers;
    // plain gzip state
    z_stream zs;
    bool zs_init = false;
    std::atomic<uint64_t> text_bytes{0};
};

// Pinned ring blocks are expensive to make (hipHostMalloc pins page by page) and a converter opens one reader per
// chromosome file: closed readers park their blocks here and the next open takes them back.
static std::mutex g_pool_mu;
static std::vector<std::pair<size_t, void *>> g_pool;
#define POOL_MAX_BLOCKS 32

static void *pool_take(size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (g_pool[i].first == bytes) {
            void *p = g_pool[i].second;
            g_pool.erase(g_pool.begin() + (long)i);
            return p;
        }
    return nullptr;
}

static bool pool_give(size_t bytes, void *p)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() >= POOL_MAX_BLOCKS) return false;
    g_pool.emplace_back(bytes, p);
    return true;
}

extern "C" void hhgt_reader_trim_pool(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &e : g_pool) hipHostFree(e.second);
    g_pool.clear();
}

extern "C" int hhgt_reader_prewarm(uint64_t block_bytes, int n_blocks)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 0;
    if (n_blocks > POOL_MAX_BLOCKS) n_blocks = POOL_MAX_BLOCKS;
    int have = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto &e : g_pool) have += e.first == (size_t)block_bytes ? 1 : 0;
    }
    while (have < n_blocks) {
        void *p = nullptr;
        if (hipHostMalloc(&p, (size_t)block_bytes, hipHostMallocDefault) != hipSuccess) break;
        if (!pool_give((size_t)block_bytes, p)) {
            hipHostFree(p);
            break;
        }
        ++have;
    }
    return have;
}

// CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool
// shows 256 hardware threads and grants 16 CPUs' worth of time: 96 inflate threads on it only fight each other)
extern "C" int hhgt_effective_cpus(void)
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int k = CPU_COUNT(&set);
        if (k > 0 && (n <= 0 || k < n)) n = k;
    }
    const char *files[] = {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"};
    for (int f = 0; f < 2; ++f) {
        FILE *fp = fopen(files[f], "r");
        if (!fp) continue;
        char a[64] = "", b[64] = "";
        const int got = fscanf(fp, "%63s %63s", a, b);
        fclose(fp);
        long long quota = -1, period = 100000;
        if (f == 0) {
            if (got >= 1 && strcmp(a, "max") != 0) quota = atoll(a);
            if (got >= 2) period = atoll(b);
        } else {
            if (got >= 1) quota = atoll(a);
            FILE *pp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (pp) {
                if (fscanf(pp, "%63s", b) == 1) period = atoll(b);
                fclose(pp);
            }
        }
        if (quota > 0 && period > 0) {
            const int k = (int)((quota + period - 1) / period);
            if (k > 0 && (n <= 0 || k < n)) n = k;
        }
        break;
    }
    return n > 0 ? n : 1;
}

static bool looks_bgzf(const uint8_t *p, size_t n)
{
    return n >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 4) && p[10] == 6 && p[11] == 0 &&
           p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0;
}

// ---- CRC-32 (RFC 1952) --------------------------------------------------------------------------------------------
// table form, sixteen bytes per step (portable path and tails)
static uint32_t g_crc_tab[16][256];
static std::once_flag g_crc_once;

static void crc_init()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        g_crc_tab[0][i] = c;
    }
    for (int t = 1; t < 16; ++t)
        for (uint32_t i = 0; i < 256; ++i) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFFu];
}

// raw update: c is the running register (already inverted), returns the register
static uint32_t crc32_table_update(uint32_t c, const uint8_t *p, size_t n)
{
    std::call_once(g_crc_once, crc_init);
    const uint32_t(*T)[256] = g_crc_tab;
    while (n >= 16) {
        uint32_t a, b, d, e;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        memcpy(&d, p + 8, 4);
        memcpy(&e, p + 12, 4);
        a ^= c;
        c = T[15][a & 0xFFu] ^ T[14][(a >> 8) & 0xFFu] ^ T[13][(a >> 16) & 0xFFu] ^ T[12][a >> 24] ^
            T[11][b & 0xFFu] ^ T[10][(b >> 8) & 0xFFu] ^ T[9][(b >> 16) & 0xFFu] ^ T[8][b >> 24] ^
            T[7][d & 0xFFu] ^ T[6][(d >> 8) & 0xFFu] ^ T[5][(d >> 16) & 0xFFu] ^ T[4][d >> 24] ^
            T[3][e & 0xFFu] ^ T[2][(e >> 8) & 0xFFu] ^ T[1][(e >> 16) & 0xFFu] ^ T[0][e >> 24];
        p += 16;
        n -= 16;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// Carry-less-multiply folding (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", the
// scheme zlib-ng / Chromium's zlib use): four 128-bit lanes are folded over the input 64 bytes per step, then
// into one lane, then reduced to 32 bits (Barrett).  Constants are x^k mod P for the bit-reflected polynomial
// 0xEDB88320.  n >= 64 and n % 16 == 0; the caller feeds the tail through the table form.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul_update(uint32_t crc, const uint8_t *buf, size_t len)
{
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596, 0x0000000154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009e, 0x00000001751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000000000, 0x0000000163cd6124);
    const __m128i poly = _mm_set_epi64x(0x00000001f7011641, 0x00000001db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64;
    len -= 64;
    while (len >= 64) {   // fold four lanes in parallel
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = k3k4;   // fold the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {   // single-lane folds over the remaining 16-byte pieces
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction 64 -> 32 bits
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

static bool have_clmul()
{
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}
#endif

extern "C" int hhgt_fast_inflate(const void *in, uint64_t in_len, void *out, uint64_t out_len)
{
    return hhgt_fast_inflate_impl(static_cast<const uint8_t *>(in), (size_t)in_len, static_cast<uint8_t *>(out), (size_t)out_len);
}

extern "C" uint32_t hhgt_crc32(const void *data, uint64_t n)
{
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint32_t c = 0xFFFFFFFFu;
#if defined(__x86_64__)
    if (n >= 64 && have_clmul()) {
        const size_t body = (size_t)n & ~(size_t)15;
        c = crc32_clmul_update(c, p, body);
        p += body;
        n -= body;
    }
#endif
    return ~crc32_table_update(c, p, (size_t)n);
}

// ---- tasks --------------------------------------------------------------------------------------------------------
static int run_task(z_stream *zs, const Task &t)
{
    if (t.fd >= 0) {
        // plain-text piece of an uncompressed file: pread straight into the pinned block.  (A memcpy out of the
        // mapping takes a minor fault per 4 KiB page it touches first; the copy inside read() does not.)
        size_t got = 0;
        while (got < t.dst_len) {
            ssize_t k = pread(t.fd, t.dst + got, t.dst_len - got, (off_t)(t.file_off + got));
            if (k <= 0) return -1;
            got += (size_t)k;
        }
        return 0;
    }
    // own decoder first (fast_inflate.h: ~4x zlib on genotype text); whatever it does not accept goes to zlib, which
    // stays the arbiter of what is a corrupt member (HHGT_ZLIB_INFLATE=1: zlib only)
    static const bool zlib_only = getenv("HHGT_ZLIB_INFLATE") && atoi(getenv("HHGT_ZLIB_INFLATE")) != 0;
    if (zlib_only || hhgt_fast_inflate_impl(t.src, t.src_len, t.dst, t.dst_len) != 0) {
        if (inflateReset(zs) != Z_OK) return -1;
        zs->next_in = const_cast<Bytef *>(t.src);
        zs->avail_in = t.src_len;
        zs->next_out = t.dst;
        zs->avail_out = t.dst_len;
        int rc = inflate(zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs->avail_out != 0) return -1;
    }
    if (t.check_crc) {
        // the member's trailer (CRC32, ISIZE) follows the payload; htslib's bgzf.c rejects a member whose text does
        // not hash to it ("CRC32 checksum mismatch")
        const uint8_t *tr = t.src + t.src_len;
        const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        if (hhgt_crc32(t.dst, t.dst_len) != want) return -2;
    }
    return 0;
}

static void set_error(hhgt_reader *r, const char *msg)
{
    std::lock_guard<std::mutex> lk(r->mu);
    if (!r->error) {
        r->error = HHGT_ERR_IO;
        r->errmsg = msg;
    }
    r->eof = true;
    r->cv_pub.notify_all();
    r->cv_open.notify_all();
    r->cv_free.notify_all();
}

#define TASK_BATCH 4u

static void worker_main(hhgt_reader *r)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    for (;;) {
        int bi;
        uint32_t gen, n_tasks;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_open.wait(lk, [&] { return r->stop || !r->open_.empty(); });
            if (r->stop) break;
            bi = r->open_.front();
            gen = r->ring[bi].gen;
            n_tasks = r->ring[bi].n_tasks;
        }
        Block &b = r->ring[bi];
        for (;;) {
            uint64_t cur = b.ticket.load();
            bool got = false;
            while ((uint32_t)(cur >> 32) == gen && (uint32_t)cur < n_tasks) {
                if (b.ticket.compare_exchange_weak(cur, cur + TASK_BATCH)) {
                    got = true;
                    break;
                }
            }
            if (!got) break;
            const uint32_t t0 = (uint32_t)cur;
            const uint32_t t1 = t0 + TASK_BATCH < n_tasks ? t0 + TASK_BATCH : n_tasks;
            for (uint32_t t = t0; t < t1; ++t) {
                const int rc = run_task(&zs, b.tasks[t]);
                if (rc != 0) set_error(r, rc == -2 ? "BGZF member: CRC32 checksum mismatch" : "inflate failed");
            }
            if (b.done.fetch_add(t1 - t0) + (t1 - t0) == n_tasks) {
                std::lock_guard<std::mutex> lk(r->mu);   // the block's last task: it may be published now
                r->cv_pub.notify_all();
            }
        }
        {
            std::lock_guard<std::mutex> lk(r->mu);       // exhausted: take it off the open list (once per life)
            if (!r->open_.empty() && r->open_.front() == bi && r->ring[bi].gen == gen) r->open_.pop_front();
        }
    }
    inflateEnd(&zs);
}

// scanner side: a free block (waits), or -1 when the reader is closing
static int take_free(hhgt_reader *r)
{
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv_free.wait(lk, [&] { return r->stop || !r->free_.empty(); });
    if (r->stop) return -1;
    int bi = r->free_.front();
    r->free_.pop_front();
    Block &b = r->ring[bi];
    b.tasks.clear();
    b.n_tasks = 0;
    b.n = 0;
    b.scanned = false;
    b.last = false;
    return bi;
}

// the block's layout is final: open it to the workers and put it in publication order
static void open_block(hhgt_reader *r, int bi, size_t pub, bool last)
{
    Block &b = r->ring[bi];
    b.n = pub;
    b.n_tasks = (uint32_t)b.tasks.size();
    b.last = last;
    r->text_bytes.fetch_add(pub);
    {
        std::lock_guard<std::mutex> lk(r->mu);
        b.gen += 1;
        b.done.store(0);
        b.ticket.store((uint64_t)b.gen << 32);
        b.scanned = true;
        r->order.push_back(bi);
        if (b.n_tasks) r->open_.push_back(bi);
        if (last) r->eof = true;
    }
    r->cv_open.notify_all();
    r->cv_pub.notify_all();
}

// ---- BGZF: barrier-free scanner --------------------------------------------------------------------------------
static void scan_bgzf(hhgt_reader *r)
{
    std::vector<uint8_t> carry;   // partial last line of the previous block
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    bool input_done = r->in_pos >= r->map_len;
    while (!input_done) {
        const int bi = take_free(r);
        if (bi < 0) break;
        Block &b = r->ring[bi];
        size_t n = carry.size();
        if (n) memcpy(b.buf, carry.data(), n);
        carry.clear();
        const size_t first_member_at = n;
        // lay out members while they fit
        while (r->in_pos < r->map_len) {
            const uint8_t *p = r->map + r->in_pos;
            const size_t left = r->map_len - r->in_pos;
            if (!looks_bgzf(p, left)) {
                set_error(r, "corrupt BGZF block header");
                inflateEnd(&zs);
                return;
            }
            const uint32_t bsize = (uint32_t)p[16] | ((uint32_t)p[17] << 8);
            const size_t total = (size_t)bsize + 1;
            if (total > left || total < 26) {
                set_error(r, "truncated compressed input");
                inflateEnd(&zs);
                return;
            }
            const uint32_t isize = (uint32_t)p[total - 4] | ((uint32_t)p[total - 3] << 8) | ((uint32_t)p[total - 2] << 16) |
                                   ((uint32_t)p[total - 1] << 24);
            if (isize > 65536) {
                set_error(r, "inflate failed");
                inflateEnd(&zs);
                return;
            }
            if (n + isize > r->block_bytes) break;
            if (isize) {
                Task t;
                t.src = p + 18;
                t.src_len = (uint32_t)(total - 18 - 8);
                t.dst = b.buf + n;
                t.dst_len = isize;
                t.check_crc = r->check_crc;
                b.tasks.push_back(t);
            }
            n += isize;
            r->in_pos += total;
        }
        input_done = r->in_pos >= r->map_len;
        if (b.tasks.empty() && !input_done) {
            set_error(r, "a line is longer than the reader's block size");
            break;
        }
        size_t pub = n;
        if (!input_done) {
            // Where does the last whole line end?  Inflate members from the back (normally just the last one) until
            // one holds a newline; they are done here, so they leave the task list.
            size_t cut = (size_t)-1;
            while (!b.tasks.empty()) {
                const Task t = b.tasks.back();
                b.tasks.pop_back();
                const int rc = run_task(&zs, t);
                if (rc != 0) {
                    set_error(r, rc == -2 ? "BGZF member: CRC32 checksum mismatch" : "inflate failed");
                    inflateEnd(&zs);
                    return;
                }
                const void *nl = memrchr(t.dst, '\n', t.dst_len);
                if (nl) {
                    cut = (size_t)((const uint8_t *)nl - b.buf) + 1;
                    break;
                }
            }
            if (cut == (size_t)-1) {
                // no newline in any member of this block: only the carried bytes may hold one
                const void *nl = first_member_at ? memrchr(b.buf, '\n', first_member_at) : nullptr;
                if (!nl) {
                    set_error(r, "a line is longer than the reader's block size");
                    break;
                }
                cut = (size_t)((const uint8_t *)nl - b.buf) + 1;
            }
            pub = cut;
            carry.assign(b.buf + cut, b.buf + n);
        }
        open_block(r, bi, pub, input_done);
    }
    inflateEnd(&zs);
    if (input_done) {
        std::lock_guard<std::mutex> lk(r->mu);
        r->eof = true;
        r->cv_pub.notify_all();
    }
}

// ---- plain gzip (one stream) and uncompressed input ------------------------------------------------------------
static long long fill_gzip(hhgt_reader *r, uint8_t *dst, size_t cap)
{
    size_t produced = 0;
    while (produced < cap) {
        if (!r->zs_init) {
            if (r->in_pos >= r->map_len) break;
            memset(&r->zs, 0, sizeof(r->zs));
            if (inflateInit2(&r->zs, 15 + 32) != Z_OK) return -1;
            r->zs_init = true;
        }
        size_t in_avail = r->map_len - r->in_pos;
        uInt chunk_in = in_avail > (1u << 30) ? (1u << 30) : (uInt)in_avail;
        size_t out_avail = cap - produced;
        uInt chunk_out = out_avail > (1u << 30) ? (1u << 30) : (uInt)out_avail;
        r->zs.next_in = const_cast<Bytef *>(r->map + r->in_pos);
        r->zs.avail_in = chunk_in;
        r->zs.next_out = dst + produced;
        r->zs.avail_out = chunk_out;
        int rc = inflate(&r->zs, Z_NO_FLUSH);
        r->in_pos += chunk_in - r->zs.avail_in;
        produced += chunk_out - r->zs.avail_out;
        if (rc == Z_STREAM_END) {  // gzip member finished; another may follow
            inflateEnd(&r->zs);
            r->zs_init = false;
            continue;
        }
        if (rc == Z_BUF_ERROR && chunk_in == 0) return -2;  // truncated file
        if (rc != Z_OK && rc != Z_BUF_ERROR) return -1;
        if (r->in_pos >= r->map_len && r->zs.avail_out != 0) return -2;  // input exhausted mid-member
    }
    return (long long)produced;
}

static void scan_stream(hhgt_reader *r)
{
    std::vector<uint8_t> carry;
    bool input_done = r->map_len == 0;
    while (!input_done) {
        const int bi = take_free(r);
        if (bi < 0) return;
        Block &b = r->ring[bi];
        size_t n = carry.size();
        if (n) memcpy(b.buf, carry.data(), n);
        carry.clear();
        size_t pub;
        if (r->is_gzip) {
            const long long got = fill_gzip(r, b.buf + n, r->block_bytes - n);
            if (got < 0) {
                set_error(r, got == -2 ? "truncated compressed input" : "inflate failed");
                return;
            }
            n += (size_t)got;
            input_done = r->in_pos >= r->map_len && !r->zs_init;
            if (got == 0 && !input_done) input_done = r->in_pos >= r->map_len;
            pub = n;
            if (!input_done) {
                const void *nl = n ? memrchr(b.buf, '\n', n) : nullptr;
                if (!nl) {
                    set_error(r, "a line is longer than the reader's block size");
                    return;
                }
                pub = (size_t)((const uint8_t *)nl - b.buf) + 1;
                carry.assign(b.buf + pub, b.buf + n);
            }
        } else {
            // uncompressed: the cut is found in the mapping itself, the bytes are pread by the workers
            const size_t avail = r->map_len - r->in_pos;
            size_t take = avail < r->block_bytes - n ? avail : r->block_bytes - n;
            input_done = take == avail;
            if (!input_done) {
                const void *nl = memrchr(r->map + r->in_pos, '\n', take);
                if (!nl) {
                    set_error(r, "a line is longer than the reader's block size");
                    return;
                }
                take = (size_t)((const uint8_t *)nl - (r->map + r->in_pos)) + 1;
            }
            const size_t piece = 4u << 20;
            for (size_t o = 0; o < take; o += piece) {
                Task t;
                t.dst = b.buf + n + o;
                t.dst_len = (uint32_t)(take - o < piece ? take - o : piece);
                t.fd = r->fd;
                t.file_off = r->in_pos + o;
                b.tasks.push_back(t);
            }
            r->in_pos += take;
            n += take;
            pub = n;
        }
        open_block(r, bi, pub, input_done);
    }
    std::lock_guard<std::mutex> lk(r->mu);
    r->eof = true;
    r->cv_pub.notify_all();
}

static void producer_main(hhgt_reader *r)
{
    if (r->is_bgzf) scan_bgzf(r);
    else scan_stream(r);
}

extern "C" int hhgt_reader_open(const char *path, uint64_t block_bytes, int n_threads, int n_blocks, hhgt_reader **out)
{
    if (!path || !out) return HHGT_ERR_ARG;
    *out = nullptr;
    if (block_bytes < (1u << 20)) block_bytes = 1u << 20;
    if (n_blocks <= 0) n_blocks = 4;
    if (n_blocks < 2) n_blocks = 2;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        hhgt_set_error("cannot open %s", path);
        return HHGT_ERR_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        hhgt_set_error("cannot stat %s", path);
        return HHGT_ERR_IO;
    }
    hhgt_reader *r = new hhgt_reader();
    r->fd = fd;
    r->map_len = (size_t)st.st_size;
    if (r->map_len) {
        void *m = mmap(nullptr, r->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            delete r;
            hhgt_set_error("cannot mmap %s", path);
            return HHGT_ERR_IO;
        }
        madvise(m, r->map_len, MADV_SEQUENTIAL);
        r->map = static_cast<const uint8_t *>(m);
    }
    r->is_gzip = r->map_len >= 2 && r->map[0] == 0x1f && r->map[1] == 0x8b;
    r->is_bgzf = r->is_gzip && looks_bgzf(r->map, r->map_len);
    {
        const char *e = getenv("HHGT_BGZF_NO_CRC");
        r->check_crc = !(e && *e && *e != '0');
    }
    r->block_bytes = (size_t)block_bytes;
    r->n_blocks = n_blocks;
    r->ring.reset(new Block[(size_t)n_blocks]);
    r->held.assign((size_t)n_blocks, 0);
    // pinned when a HIP device exists (the copy engine can then DMA straight out of the ring);
    // plain pages otherwise (CPU-only hosts still frame and inflate, e.g. in the CPU test suite)
    int ndev = 0;
    r->pinned = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    for (int i = 0; i < n_blocks; ++i) {
        void *p = r->pinned ? pool_take(r->block_bytes) : nullptr;
        if (!p && r->pinned && hipHostMalloc(&p, r->block_bytes, hipHostMallocDefault) != hipSuccess) {
            r->pinned = false;   // (blocks already taken stay pinned: pinned and pageable memory are freed alike below)
            p = nullptr;
        }
        r->ring[i].pinned = p != nullptr;
        if (!p) p = malloc(r->block_bytes);
        if (!p) {
            hhgt_set_error("reader: out of memory for %zu-byte blocks", r->block_bytes);
            hhgt_reader_close(r);
            return HHGT_ERR_IO;
        }
        r->ring[i].buf = static_cast<uint8_t *>(p);
        r->free_.push_back(i);
    }
    if (r->is_bgzf || !r->is_gzip) {
        // BGZF default: one worker per CPU the process may use (hhgt_effective_cpus), at most 96; HHGT_READER_THREADS
        // or the n_threads argument override.  Uncompressed input: 16 pread threads saturate the page cache copy.
        const char *e = getenv("HHGT_READER_THREADS");
        const int eff = hhgt_effective_cpus();
        int nt = n_threads > 0 ? n_threads : (e && atoi(e) > 0 ? atoi(e) : (eff < 96 ? eff : 96));
        if (nt > 192) nt = 192;
        if (!r->is_bgzf && nt > 16) nt = 16;
        for (int i = 0; i < nt; ++i) r->workers.emplace_back(worker_main, r);
    }
    r->producer = std::thread(producer_main, r);
    *out = r;
    return HHGT_OK;
}

extern "C" void hhgt_reader_close(hhgt_reader *r)
{
    if (!r) return;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->stop = true;
    }
    r->cv_free.notify_all();
    r->cv_open.notify_all();
    r->cv_pub.notify_all();
    if (r->producer.joinable()) r->producer.join();
    for (auto &t : r->workers) t.join();
    if (r->zs_init) inflateEnd(&r->zs);
    for (int i = 0; i < r->n_blocks; ++i) {
        uint8_t *b = r->ring[i].buf;
        if (!b) continue;
        if (r->ring[i].pinned) {
            if (!pool_give(r->block_bytes, b)) hipHostFree(b);
        } else
            free(b);
    }
    if (r->map) munmap(const_cast<uint8_t *>(r->map), r->map_len);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

extern "C" int hhgt_reader_is_bgzf(const hhgt_reader *r) { return r && r->is_bgzf ? 1 : 0; }

extern "C" int hhgt_reader_acquire(hhgt_reader *r, const void **host_ptr, uint64_t *nbytes, int *token, int *is_last)
{
    if (!r || !host_ptr || !nbytes || !token) return HHGT_ERR_ARG;
    *host_ptr = nullptr;
    *nbytes = 0;
    *token = -1;
    if (is_last) *is_last = 0;
    std::unique_lock<std::mutex> lk(r->mu);
    for (;;) {
        if (r->error) {
            hhgt_set_error("reader: %s", r->errmsg.c_str());
            return r->error;
        }
        if (!r->order.empty()) {
            Block &b = r->ring[r->order.front()];
            if (b.scanned && b.done.load() >= b.n_tasks) break;
        } else if (r->eof) {
            return HHGT_OK;  // end of file
        }
        if (r->stop) return HHGT_OK;
        r->cv_pub.wait(lk);
    }
    const int bi = r->order.front();
    r->order.pop_front();
    Block &b = r->ring[bi];
    if (b.n == 0) {   // an empty block (e.g. only empty members): recycle and look again
        r->free_.push_back(bi);
        r->cv_free.notify_all();
        const bool last = b.last;
        lk.unlock();
        if (last) return HHGT_OK;
        return hhgt_reader_acquire(r, host_ptr, nbytes, token, is_last);
    }
    r->held[(size_t)bi] = 1;
    *host_ptr = b.buf;
    *nbytes = b.n;
    *token = bi;
    if (is_last) *is_last = b.last ? 1 : 0;
    return HHGT_OK;
}

extern "C" int hhgt_reader_release(hhgt_reader *r, int token)
{
    if (!r || token < 0 || token >= r->n_blocks) return HHGT_ERR_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    if (!r->held[(size_t)token]) return HHGT_ERR_ARG;
    r->held[(size_t)token] = 0;
    r->free_.push_back(token);
    r->cv_free.notify_all();
    return HHGT_OK;
}

extern "C" int hhgt_reader_next(hhgt_reader *r, const void **host_ptr, uint64_t *nbytes)
{
    if (!r || !host_ptr || !nbytes) return HHGT_ERR_ARG;
    if (r->auto_held >= 0) {
        hhgt_reader_release(r, r->auto_held);
        r->auto_held = -1;
    }
    int token = -1;
    const int rc = hhgt_reader_acquire(r, host_ptr, nbytes, &token, nullptr);
    if (rc == HHGT_OK) r->auto_held = token;
    return rc;
}

extern "C" int hhgt_reader_copy_async(hhgt_reader *r, const void *host_ptr, uint64_t nbytes, void *d_dst, void *stream)
{
    if (!r || !host_ptr || !d_dst) return HHGT_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(d_dst, host_ptr, nbytes, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)));
    return HHGT_OK;
}

extern "C" int hhgt_reader_stats(const hhgt_reader *r, uint64_t *file_bytes, uint64_t *text_bytes)
{
    if (!r) return HHGT_ERR_ARG;
    if (file_bytes) *file_bytes = r->in_pos;
    if (text_bytes) *text_bytes = r->text_bytes.load();
    return HHGT_OK;
}

// ---- bench / test tooling: parallel BGZF writer (declared in include/hhgt_synth.h) ---------------------------------
#include "../../include/hhgt_synth.h"
#include <stdio.h>

static bool bgzf_member(z_stream *zs, int level, const uint8_t *src, uint32_t n, std::vector<uint8_t> &out)
{
    const size_t at = out.size();
    out.resize(at + 18 + 65536 + 8);
    uint8_t *h = out.data() + at;
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(h, head, 16);
    (void)level;
    if (deflateReset(zs) != Z_OK) return false;
    zs->next_in = const_cast<Bytef *>(src);
    zs->avail_in = n;
    zs->next_out = h + 18;
    zs->avail_out = 65536 - 18 - 8;
    if (deflate(zs, Z_FINISH) != Z_STREAM_END) return false;
    const uint32_t clen = (uint32_t)(65536 - 18 - 8 - zs->avail_out);
    const uint32_t bsize = clen + 25;   // total member size - 1
    h[16] = (uint8_t)(bsize & 0xFF);
    h[17] = (uint8_t)(bsize >> 8);
    uint8_t *t = h + 18 + clen;
    const uint32_t crc = hhgt_crc32(src, n);
    for (int k = 0; k < 4; ++k) t[k] = (uint8_t)(crc >> (8 * k));
    for (int k = 0; k < 4; ++k) t[4 + k] = (uint8_t)(n >> (8 * k));
    out.resize(at + 18 + clen + 8);
    return true;
}

extern "C" int hhgt_synth_write_bgzf(const char *path, const void *text, uint64_t nbytes, int level, int n_threads)
{
    if (!path || (!text && nbytes)) return HHGT_ERR_ARG;
    const uint32_t piece = 0xFF00;
    const uint64_t n_members = (nbytes + piece - 1) / piece;
    unsigned hw = std::thread::hardware_concurrency();
    int nt = n_threads > 0 ? n_threads : (int)(hw ? hw : 4);
    if ((uint64_t)nt > n_members) nt = (int)(n_members ? n_members : 1);
    std::vector<std::vector<uint8_t>> parts((size_t)nt);
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    const uint8_t *src = static_cast<const uint8_t *>(text);
    for (int i = 0; i < nt; ++i) {
        th.emplace_back([&, i] {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                bad.store(1);
                return;
            }
            const uint64_t m0 = n_members * (uint64_t)i / (uint64_t)nt, m1 = n_members * (uint64_t)(i + 1) / (uint64_t)nt;
            parts[(size_t)i].reserve((size_t)((m1 - m0) * 4096));
            for (uint64_t m = m0; m < m1; ++m) {
                const uint64_t o = m * piece;
                const uint32_t n = (uint32_t)(nbytes - o < piece ? nbytes - o : piece);
                if (!bgzf_member(&zs, level, src + o, n, parts[(size_t)i])) {
                    bad.store(1);   // (incompressible input would need a smaller member; the generator's text never is)
                    break;
                }
            }
            deflateEnd(&zs);
        });
    }
    for (auto &t : th) t.join();
    if (bad.load()) {
        hhgt_set_error("write_bgzf: deflate failed");
        return HHGT_ERR_IO;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        hhgt_set_error("cannot create %s", path);
        return HHGT_ERR_IO;
    }
    bool ok = true;
    for (auto &p : parts) ok = ok && (p.empty() || fwrite(p.data(), 1, p.size(), f) == p.size());
    // empty end-of-file member, as bgzip writes it (deflate of nothing = 03 00)
    static const uint8_t eof_member[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0,
                                           0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_member, 1, 28, f) == 28;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        hhgt_set_error("short write to %s", path);
        return HHGT_ERR_IO;
    }
    return HHGT_OK;
}
