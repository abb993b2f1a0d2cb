// reader.hip — host-side VCF text reader (see include/hhgt_reader.h): mmap + zlib inflate (BGZF blocks
// in parallel, plain gzip streaming) into a ring of pinned, line-aligned blocks; hipMemcpyAsync to HBM.
// Host C++ only (compiled by hipcc with the rest of libhhgt.so); no device code in this file.
#include "common.h"
#include "../../include/hhgt_reader.h"
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace {

struct Task {
    const uint8_t *src;
    uint32_t src_len;
    uint8_t *dst;
    uint32_t dst_len;
    int fd = -1;            // >= 0: an uncompressed piece, read with pread(fd, dst, dst_len, file_off)
    uint64_t file_off = 0;
    bool check_crc = false;  // BGZF member: compare crc32(dst) with the trailer behind src
};

struct Block {
    uint8_t *buf = nullptr;
    size_t n = 0;
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
    std::vector<Block> ring;
    std::deque<int> filled, free_;
    int held = -1;
    bool eof = false, stop = false;
    int error = 0;
    std::string errmsg;
    std::mutex mu;
    std::condition_variable cv;
    std::thread producer;
    // BGZF worker pool
    std::vector<std::thread> workers;
    std::vector<Task> tasks;
    // ticket = generation << 32 | next task index: a worker that draws a ticket of a finished
    // generation can never touch the task list of the next one
    std::atomic<uint64_t> ticket{0};
    std::atomic<uint64_t> open_gen{0};
    std::atomic<size_t> n_tasks{0};
    std::atomic<size_t> done_tasks{0};
    std::atomic<int> task_err{0};
    uint64_t batch_id = 0;
    bool pool_stop = false;
    std::mutex pmu;
    std::condition_variable pcv, dcv;
    // plain gzip state
    z_stream zs;
    bool zs_init = false;
    std::atomic<uint64_t> text_bytes{0};
};

static bool looks_bgzf(const uint8_t *p, size_t n)
{
    return n >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 4) && p[10] == 6 && p[11] == 0 &&
           p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0;
}

// CRC-32 (RFC 1952) sixteen bytes per step: zlib 1.2.11's crc32() in this image runs near 1 GB/s per core, half of
// what its inflate delivers on VCF text, and would make the check cost as much as the inflate it guards.
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

static uint32_t crc32_fast(const uint8_t *p, size_t n)
{
    std::call_once(g_crc_once, crc_init);
    uint32_t c = 0xFFFFFFFFu;
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
    return ~c;
}

static int inflate_raw(z_stream *zs, const Task &t)
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
    if (inflateReset(zs) != Z_OK) return -1;
    zs->next_in = const_cast<Bytef *>(t.src);
    zs->avail_in = t.src_len;
    zs->next_out = t.dst;
    zs->avail_out = t.dst_len;
    int rc = inflate(zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs->avail_out != 0) return -1;
    if (t.check_crc) {
        // the member's trailer (CRC32, ISIZE) follows the payload; htslib's bgzf.c rejects a member whose text does
        // not hash to it ("CRC32 checksum mismatch")
        const uint8_t *tr = t.src + t.src_len;
        const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        if (crc32_fast(t.dst, t.dst_len) != want) return -2;
    }
    return 0;
}

static void run_tasks(hhgt_reader *r, z_stream *zs)
{
    for (;;) {
        const uint64_t t = r->ticket.fetch_add(1);
        const uint64_t gen = t >> 32, i = t & 0xFFFFFFFFull;
        if (gen != r->open_gen.load() || i >= r->n_tasks.load()) break;
        const int rc = inflate_raw(zs, r->tasks[i]);
        if (rc != 0) r->task_err.store(rc == -2 ? 2 : 1);
        r->done_tasks.fetch_add(1);
    }
}

static void worker_main(hhgt_reader *r)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(r->pmu);
            r->pcv.wait(lk, [&] { return r->pool_stop || r->batch_id != seen; });
            if (r->pool_stop) break;
            seen = r->batch_id;
        }
        run_tasks(r, &zs);
        {
            std::lock_guard<std::mutex> lk(r->pmu);
        }
        r->dcv.notify_all();
    }
    inflateEnd(&zs);
}

// fills dst[0, cap) with decompressed bytes; returns bytes produced (0 = end of input), <0 on error
static long long run_batch(hhgt_reader *r);

// uncompressed input: the copy into the pinned ring is split over the worker pool (one thread tops out
// near 10 GB/s, well under what the PCIe link takes)
static long long fill_plain(hhgt_reader *r, uint8_t *dst, size_t cap)
{
    size_t avail = r->map_len - r->in_pos;
    size_t n = avail < cap ? avail : cap;
    const size_t piece = 4u << 20;
    if (n <= piece || r->workers.empty()) {
        memcpy(dst, r->map + r->in_pos, n);
    } else {
        r->n_tasks.store(0);
        r->tasks.clear();
        for (size_t o = 0; o < n; o += piece) {
            size_t l = n - o < piece ? n - o : piece;
            Task t{r->map + r->in_pos + o, (uint32_t)l, dst + o, (uint32_t)l};
            t.fd = r->fd;
            t.file_off = r->in_pos + o;
            r->tasks.push_back(t);
        }
        if (run_batch(r) < 0) return -1;
    }
    r->in_pos += n;
    return (long long)n;
}

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

static long long fill_bgzf(hhgt_reader *r, uint8_t *dst, size_t cap)
{
    r->n_tasks.store(0);  // closes the previous generation before the list is rewritten
    r->tasks.clear();
    size_t produced = 0;
    while (r->in_pos < r->map_len) {
        const uint8_t *p = r->map + r->in_pos;
        size_t left = r->map_len - r->in_pos;
        if (!looks_bgzf(p, left)) return -3;
        uint32_t bsize = (uint32_t)p[16] | ((uint32_t)p[17] << 8);
        size_t total = (size_t)bsize + 1;
        if (total > left || total < 26) return -2;
        uint32_t isize = (uint32_t)p[total - 4] | ((uint32_t)p[total - 3] << 8) | ((uint32_t)p[total - 2] << 16) |
                         ((uint32_t)p[total - 1] << 24);
        if (isize > 65536) return -1;
        if (produced + isize > cap) break;
        if (isize) {
            Task t{p + 18, (uint32_t)(total - 18 - 8), dst + produced, isize};
            t.check_crc = r->check_crc;
            r->tasks.push_back(t);
        }
        produced += isize;
        r->in_pos += total;
    }
    const long long rb = run_batch(r);
    if (rb < 0) return rb;
    return (long long)produced;
}

// runs r->tasks on the worker pool (the calling producer thread helps); <0 on failure
static long long run_batch(hhgt_reader *r)
{
    if (r->tasks.empty()) return 0;
    r->done_tasks.store(0);
    {
        std::lock_guard<std::mutex> lk(r->pmu);
        r->batch_id++;
        r->open_gen.store(r->batch_id);
        r->n_tasks.store(r->tasks.size());
        r->ticket.store(r->batch_id << 32);
    }
    r->pcv.notify_all();
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    run_tasks(r, &zs);
    inflateEnd(&zs);
    {
        std::unique_lock<std::mutex> lk(r->pmu);
        r->dcv.wait(lk, [&] { return r->done_tasks.load() >= r->tasks.size(); });
    }
    return r->task_err.load() == 2 ? -4 : (r->task_err.load() ? -1 : 0);
}

static void producer_main(hhgt_reader *r)
{
    std::vector<uint8_t> carry;
    bool input_done = false;
    while (!input_done) {
        int bi;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [&] { return r->stop || !r->free_.empty(); });
            if (r->stop) return;
            bi = r->free_.front();
            r->free_.pop_front();
        }
        Block &b = r->ring[bi];
        size_t n = carry.size();
        if (n) memcpy(b.buf, carry.data(), n);
        carry.clear();
        // fill until the block is (nearly) full or the input ends
        for (;;) {
            size_t cap = r->block_bytes - n;
            if (cap < 65536 + 1) break;
            long long got = r->is_bgzf ? fill_bgzf(r, b.buf + n, cap)
                            : r->is_gzip ? fill_gzip(r, b.buf + n, cap)
                                         : fill_plain(r, b.buf + n, cap);
            if (got < 0) {
                std::lock_guard<std::mutex> lk(r->mu);
                r->error = HHGT_ERR_IO;
                r->errmsg = got == -2 ? "truncated compressed input" : got == -3 ? "corrupt BGZF block header" : got == -4 ? "BGZF member: CRC32 checksum mismatch" : "inflate failed";
                r->eof = true;
                r->cv.notify_all();
                return;
            }
            if (got == 0) {
                if (r->in_pos >= r->map_len) input_done = true;
                break;
            }
            n += (size_t)got;
            if (!r->is_bgzf) break;  // gzip/plain fill the whole capacity in one call
        }
        if (!r->is_bgzf && r->in_pos >= r->map_len && !r->zs_init) input_done = true;
        size_t pub = n;
        if (!input_done) {
            // cut at the last newline; the partial line moves to the next block
            const void *nl = n ? memrchr(b.buf, '\n', n) : nullptr;
            if (!nl) {
                if (n + 65536 + 1 > r->block_bytes) {
                    std::lock_guard<std::mutex> lk(r->mu);
                    r->error = HHGT_ERR_IO;
                    r->errmsg = "a line is longer than the reader's block size";
                    r->eof = true;
                    r->cv.notify_all();
                    return;
                }
                pub = 0;
                carry.assign(b.buf, b.buf + n);
            } else {
                pub = (size_t)((const uint8_t *)nl - b.buf) + 1;
                carry.assign(b.buf + pub, b.buf + n);
            }
        }
        b.n = pub;
        r->text_bytes.fetch_add(pub);
        {
            std::lock_guard<std::mutex> lk(r->mu);
            if (pub) r->filled.push_back(bi);
            else r->free_.push_back(bi);
            if (input_done) r->eof = true;
        }
        r->cv.notify_all();
    }
}

extern "C" int hhgt_reader_open(const char *path, uint64_t block_bytes, int n_threads, int n_blocks, hhgt_reader **out)
{
    if (!path || !out) return HHGT_ERR_ARG;
    *out = nullptr;
    if (block_bytes < (1u << 20)) block_bytes = 1u << 20;
    if (n_blocks <= 0) n_blocks = 3;
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
    r->ring.resize((size_t)n_blocks);
    // pinned when a HIP device exists (the copy engine can then DMA straight out of the ring);
    // plain pages otherwise (CPU-only hosts still frame and inflate, e.g. in the CPU test suite)
    int ndev = 0;
    r->pinned = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    for (auto &b : r->ring) {
        void *p = nullptr;
        if (r->pinned && hipHostMalloc(&p, r->block_bytes, hipHostMallocDefault) != hipSuccess) {
            r->pinned = false;
            p = nullptr;
        }
        if (!p) p = malloc(r->block_bytes);
        if (!p) {
            hhgt_set_error("reader: out of memory for %zu-byte blocks", r->block_bytes);
            hhgt_reader_close(r);
            return HHGT_ERR_IO;
        }
        b.buf = static_cast<uint8_t *>(p);
    }
    for (int i = 0; i < n_blocks; ++i) r->free_.push_back(i);
    if (r->is_bgzf || !r->is_gzip) {
        unsigned hw = std::thread::hardware_concurrency();
        // measured on the 256-thread host of an MI355X box: 32-64 inflate threads saturate (~20 GB/s of text);
        // more threads lose to wake-up and memory contention
        int nt = n_threads > 0 ? n_threads : (hw ? (int)(hw < 48 ? hw : 48) : 4);
        if (nt > 192) nt = 192;
        if (!r->is_bgzf && nt > 16) nt = 16;  // plain pread copies: 16 threads 28 GB/s, 32 threads 25 GB/s (contention)
        for (int i = 0; i < nt - 1; ++i) r->workers.emplace_back(worker_main, r);
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
    r->cv.notify_all();
    if (r->producer.joinable()) r->producer.join();
    {
        std::lock_guard<std::mutex> lk(r->pmu);
        r->pool_stop = true;
    }
    r->pcv.notify_all();
    for (auto &t : r->workers) t.join();
    if (r->zs_init) inflateEnd(&r->zs);
    for (auto &b : r->ring) {
        if (!b.buf) continue;
        if (r->pinned) hipHostFree(b.buf);
        else free(b.buf);
    }
    if (r->map) munmap(const_cast<uint8_t *>(r->map), r->map_len);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

extern "C" int hhgt_reader_is_bgzf(const hhgt_reader *r) { return r && r->is_bgzf ? 1 : 0; }

extern "C" int hhgt_reader_next(hhgt_reader *r, const void **host_ptr, uint64_t *nbytes)
{
    if (!r || !host_ptr || !nbytes) return HHGT_ERR_ARG;
    *host_ptr = nullptr;
    *nbytes = 0;
    std::unique_lock<std::mutex> lk(r->mu);
    if (r->held >= 0) {
        r->free_.push_back(r->held);
        r->held = -1;
        r->cv.notify_all();
    }
    r->cv.wait(lk, [&] { return !r->filled.empty() || r->eof; });
    if (r->filled.empty()) {
        if (r->error) {
            hhgt_set_error("reader: %s", r->errmsg.c_str());
            return r->error;
        }
        return HHGT_OK;  // end of file
    }
    r->held = r->filled.front();
    r->filled.pop_front();
    *host_ptr = r->ring[r->held].buf;
    *nbytes = r->ring[r->held].n;
    return HHGT_OK;
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
