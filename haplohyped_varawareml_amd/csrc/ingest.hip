// ingest.hip — the streaming ingest engine (include/hhgt_ingest.h): files / host text -> framed genotype chunks on
// the host.  Host C++ only (threads, streams, events); every computation is one of libhhgt's kernels, reached through
// the same C entry points a caller would use (hhgt_encode_text_async, hhgt_pad_tail_cursor, hhgt_compress_chunks) or,
// for the device-side BGZF inflate issued from the source thread, through launch_inflate directly.
//
// Why it looks the way it does (round 1's Python loop reached 1.6 / 4.3 M variants/s, host / device inflate, against
// 58 M/s for text already in HBM): every text block cost the host a stream synchronisation for the line count, one for
// the kept count, one for the chunk sizes, pageable uploads of the member tables and a Python callback in between.
// Here the GPU's queue never drains: the driver thread queues block k's encode BEFORE it looks at block k-1's result
// record, the compressed sizes of a batch are read by another thread one stage later, and all staging memory is
// pinned and reused.
#include "common.h"
#include "../../include/hhgt_ingest.h"
#include "../../include/hhgt_reader.h"
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// blocking FIFO; close() wakes everybody, pop() then drains what is left and returns false
template <class T> struct BQ {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(v));
        }
        cv.notify_all();
    }
    bool pop(T &out)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return closed || !q.empty(); });
        if (q.empty()) return false;
        out = std::move(q.front());
        q.pop_front();
        return true;
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            closed = true;
        }
        cv.notify_all();
    }
    size_t size()
    {
        std::lock_guard<std::mutex> lk(m);
        return q.size();
    }
};

struct PinnedBuf {   // grows on demand; pinned allocations are slow, so they only ever grow
    uint8_t *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return HHGT_OK;
        if (p) hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 4096;
        static const bool dbg = getenv("HHGT_ALLOC_DEBUG") != nullptr;
        if (dbg)
            fprintf(stderr, "[alloc] %.3f ms  hipHostMalloc %zu\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e3, want);
        if (hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            hhgt_set_error("ingest: hipHostMalloc(%zu) failed", want);
            return HHGT_ERR_HIP;
        }
        cap = want;
        return HHGT_OK;
    }
    void release()
    {
        if (p) hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Input {
    int index = 0;
    int kind = 0;   // 0 file, 1 memory
    std::string path, region;
    const uint8_t *mem = nullptr;
    uint64_t mem_bytes = 0;
    // discovered by the source thread
    std::string header;
    uint64_t S = 0, S_file = 0, header_lines = 0;
    bool dev_inflate = false, is_bgzf = false;
    hhgt_reader *rd = nullptr;   // opened (possibly ahead of time) by the source thread
    // written by the driver / shipper
    hhgt_ingest_stats st;
    double t_first = 0;
    std::string last_run;
    void *state = nullptr;       // hhgt_ingest::InState of this input (driver thread)
    bool header_sent = false;
    Input() { memset(&st, 0, sizeof(st)); }
};

struct TextBuf {   // device-resident text block
    uint8_t *d = nullptr;
    size_t cap = 0;
    hipEvent_t ready = nullptr, carry_done = nullptr;
    uint64_t nbytes = 0;
    Input *in = nullptr;
    bool first = false, last = false, end_of_inputs = false;
    // device-side inflate: per-member status and the count of failures (checked at harvest)
    DevBuf status, bad;
    unsigned long long *h_bad = nullptr;   // pinned
    uint64_t n_members = 0, first_member = 0;
};

struct Staging {   // device-inflate: one block's compressed bytes + member tables, host (pinned) and device
    PinnedBuf h;
    DevBuf d;
    hipEvent_t done = nullptr;   // the inflate that read the device copy has finished
    bool used = false;
};

enum { B_VARIANTS = 2, B_COLUMNS = 3, B_INPUT_END = 4, B_HEADER = 1, B_END = 0 };

struct Batch {
    int kind = 0;
    Input *in = nullptr;
    hipEvent_t ev = nullptr;   // recorded on the main stream behind the batch's device work
    // VARIANTS
    int var_slot = -1;
    uint64_t first_variant = 0, n_variants = 0;
    uint32_t n_runs = 0;
    // COLUMNS
    int dst_slot = -1, out_slot = -1;
    uint64_t n_chunks = 0, first_col = 0, n_cols = 0, raw_bytes = 0, framed_bytes = 0;
};

struct VarSlot {
    PinnedBuf start, ref, alt;
    uint64_t run_first[HHGT_RESULT_RUNS];
    char run_names[HHGT_RESULT_RUNS][32];
};
struct DstSlot {
    DevBuf d, off;          // framed bytes, chunk_off (device)
    PinnedBuf h_off;        // chunk_off (host)
};
struct OutSlot {
    PinnedBuf h;            // framed bytes (host)
    std::vector<uint64_t> off;
};

#define N_TEXT_HOST 4
#define N_TEXT_DEV 4
#define N_RES 4
#define N_VAR 6
#define N_DST 4
#define N_OUT 5
#define N_STG 4   // device-inflate staging slots: one per text buffer, so the source never waits for an inflate two blocks back

}  // namespace

struct hhgt_ingest {
    hhgt_ctx *ctx = nullptr;
    hhgt_ingest_opts o;
    int device = 0;
    hipStream_t s_main = nullptr, s_copy = nullptr, s_inf = nullptr, s_out = nullptr;
    // device inflate: the blocks alternate between s_inf and s_inf2 (one wave per member is latency-bound: the tail round of
    // one block's launch overlaps the next block's), the few bytes behind a block's last newline move on s_carry
    hipStream_t s_inf2 = nullptr, s_carry = nullptr;
    uint64_t inf_blocks = 0;              // source thread only
    hipEvent_t last_carry = nullptr;      // the latest carry copy queued (an event of some TextBuf), or null
    // inputs
    std::mutex in_mu;
    std::condition_variable in_cv;
    std::deque<std::unique_ptr<Input>> inputs;   // all inputs ever added (stable addresses)
    size_t next_input = 0;                       // source thread's position
    bool finished = false;
    // error state
    std::mutex err_mu;
    int err = 0;
    std::string errmsg;
    std::atomic<bool> failed{false};
    // text buffers
    std::vector<TextBuf> text;
    BQ<int> free_text;
    BQ<int> q_text;          // indices in order; -1 = end of inputs
    // device-inflate staging
    Staging stg[N_STG];
    int stg_next = 0;            // source thread only: the slots go round across inputs
    // member table of the stretch being cut (source thread only).  Raw arrays that only ever grow: a std::vector sized for the
    // worst case (a member per 26 bytes: a million entries for a 25 MB stretch) zero-fills 20 MB per block — 2-5 ms of the
    // source thread per 3 ms of device work (HHGT_INGEST_DEBUG=2: read_done -> scan_done)
    struct MemberTab {
        std::unique_ptr<uint64_t[]> c_off;
        std::unique_ptr<uint32_t[]> c_len, isz, crc;
        size_t cap = 0;
        void ensure(size_t n)
        {
            if (n <= cap) return;
            cap = n + n / 4;
            c_off.reset(new uint64_t[cap]);
            c_len.reset(new uint32_t[cap]);
            isz.reset(new uint32_t[cap]);
            crc.reset(new uint32_t[cap]);
        }
    } mtab;
    DevBuf crc_x2n;
    // encode state of an input (driver thread only).  Two sets, used alternately: the first blocks of input k+1 are
    // encoded while the last blocks of input k are still being harvested, so the GPU's queue does not drain at a
    // file boundary
    struct InState {
        hhgt_layout lay;
        DevBuf G, P, t_start, t_ref, t_alt, cursor;
        bool planes = false;   // the ring holds the matrix as bit planes (include/hhgt.h "Bit-plane form"); G only backs calls beyond 0 / 1 / missing
        uint64_t ring_cols = 0, col_bytes = 0, n_sc = 0, chunk_nbytes = 0, kept_per_block = 0, done_cols = 0, host_cursor = 0;
    } ist[2];
    uint64_t n_begun = 0;
    struct Res {
        hhgt_encode_result *rec = nullptr;   // pinned
        hipEvent_t ev = nullptr;
        int text_idx = -1;
        InState *st = nullptr;
    } res[N_RES];
    // slot pools
    VarSlot var[N_VAR];
    DstSlot dst[N_DST];
    OutSlot out[N_OUT];
    BQ<int> free_var, free_dst, free_out;
    std::vector<hipEvent_t> batch_events;
    BQ<hipEvent_t> free_ev;
    // stages
    BQ<Batch> q_ship, q_out;
    std::thread th_source, th_driver, th_ship;
    // where the threads spend their time (seconds; HHGT_INGEST_DEBUG=1 prints them at close)
    struct Times {
        double src_wait_text = 0, src_work = 0, drv_wait_text = 0, drv_launch = 0, drv_harvest_wait = 0, drv_harvest = 0,
               drv_begin = 0, ship_wait_ev = 0, ship_copy = 0, ship_wait_out = 0, t_open = 0, t_first_text = 0, t_end = 0;
    } tm;
    // consumer side: what the previous hhgt_ingest_next handed out
    Batch held;
    bool have_held = false;
    bool ended = false;
};

namespace {

struct TraceRec {
    double t;
    const char *tag;
    long long a, b;
};
std::mutex g_trace_mu;
std::vector<TraceRec> g_trace;
int trace_level()
{
    static const int lv = getenv("HHGT_INGEST_DEBUG") ? atoi(getenv("HHGT_INGEST_DEBUG")) : 0;
    return lv;
}
// events a host thread waits on: the waiter sleeps (hipEventBlockingSync) instead of spinning — the driver spends ~90 % of a
// host-inflated pass inside such a wait and the shipper and the source most of the rest: up to three of the (16 granted) CPUs
// the reader's inflate threads are short of.  HHGT_INGEST_SPIN=1 restores the spinning waits.
unsigned wait_event_flags()
{
    static const bool spin = getenv("HHGT_INGEST_SPIN") && atoi(getenv("HHGT_INGEST_SPIN")) != 0;
    return hipEventDisableTiming | (spin ? 0u : (unsigned)hipEventBlockingSync);
}
void trace(const char *tag, long long a = 0, long long b = 0)
{
    if (trace_level() < 2) return;
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace.push_back({now_s(), tag, a, b});
}

void fail(hhgt_ingest *g, int code, const char *msg)
{
    {
        std::lock_guard<std::mutex> lk(g->err_mu);
        if (!g->err) {
            g->err = code ? code : HHGT_ERR_IO;
            g->errmsg = msg ? msg : "ingest failed";
        }
    }
    g->failed.store(true);
    g->free_text.close();
    g->q_text.close();
    g->free_var.close();
    g->free_dst.close();
    g->free_out.close();
    g->free_ev.close();
    g->q_ship.close();
    g->q_out.close();
    g->in_cv.notify_all();
}

#define G_TRY(expr)                                  \
    do {                                             \
        int _rc = (expr);                            \
        if (_rc != HHGT_OK) {                        \
            fail(g, _rc, hhgt_last_error());         \
            return false;                            \
        }                                            \
    } while (0)
#define G_HIP(expr)                                                                                 \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            hhgt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            fail(g, HHGT_ERR_HIP, hhgt_last_error());                                               \
            return false;                                                                           \
        }                                                                                           \
    } while (0)

// '#' lines at the start of `p`: bytes, line count, number of sample columns.  false: no complete header in [p, p+n)
bool parse_header_text(const uint8_t *p, size_t n, size_t *header_bytes, uint64_t *n_samples, bool at_eof)
{
    size_t pos = 0;
    bool have = false;
    uint64_t S = 0;
    while (pos < n && p[pos] == '#') {
        const void *nl = memchr(p + pos, '\n', n - pos);
        if (!nl && !at_eof) return false;
        const size_t end = nl ? (size_t)((const uint8_t *)nl - p) : n;
        if (end - pos >= 6 && memcmp(p + pos, "#CHROM", 6) == 0) {
            size_t tabs = 0;
            size_t e = end;
            if (e > pos && p[e - 1] == '\r') --e;
            for (size_t i = pos; i < e; ++i) tabs += p[i] == '\t';
            S = tabs >= 9 ? tabs - 8 : 0;
            have = true;
        }
        pos = nl ? end + 1 : n;
    }
    if (pos >= n && !at_eof) return false;   // the header may continue in the next block
    if (!have) return false;
    *header_bytes = pos;
    *n_samples = S;
    return true;
}

bool is_bgzf_file(const char *path)
{
    uint8_t h[18];
    int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    const ssize_t k = read(fd, h, 18);
    close(fd);
    return k == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
}

// does this file take the device inflater?  mode 0: never; 1: every BGZF file; 2 ("auto"): a BGZF file whose first member
// inflates to at least twice its size — then the compressed members are the smaller load for the host-device link, and
// the link (not the inflate) is what bounds the host path at cohort widths (text crosses it at 57 GB/s = 5.7 M variants/s
// for 2504 samples; the same files through the device inflater: 11 M/s).  Other files have nothing to gain.
bool wants_device_inflate(int mode, const char *path)
{
    if (mode == 0 || !is_bgzf_file(path)) return false;
    if (mode == 1) return true;
    uint8_t h[18];
    int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    bool dev = false;
    if (read(fd, h, 18) == 18) {
        const uint32_t bsize = (uint32_t)h[16] + ((uint32_t)h[17] << 8) + 1u;   // whole member
        uint8_t t[4];
        if (bsize >= 26 && pread(fd, t, 4, (off_t)bsize - 4) == 4) {
            const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            dev = isize >= 2u * bsize;
        }
    }
    close(fd);
    return dev;
}

// ---------------------------------------------------------------------------------------------------------------
// source thread
// ---------------------------------------------------------------------------------------------------------------
bool push_text(hhgt_ingest *g, int ti)
{
    trace("src:push_text", ti, (long long)g->text[(size_t)ti].nbytes);
    g->q_text.push(ti);
    return !g->failed.load();
}

bool take_text(hhgt_ingest *g, int *ti, size_t need)
{
    const double t0 = now_s();
    const bool got = g->free_text.pop(*ti);
    g->tm.src_wait_text += now_s() - t0;
    if (!got) return false;
    TextBuf &tb = g->text[(size_t)*ti];
    if (tb.cap < need) {   // only when a caller's block is larger than the configured size
        if (tb.d) hipFree(tb.d);
        tb.d = nullptr;
        tb.cap = 0;
        G_HIP(hipMalloc(reinterpret_cast<void **>(&tb.d), need + 256));
        tb.cap = need + 256;
    }
    tb.first = tb.last = tb.end_of_inputs = false;
    tb.n_members = 0;
    return true;
}

bool open_reader(hhgt_ingest *g, Input *in)
{
    if (in->rd || in->kind != 0) return true;
    const uint64_t bb = g->o.block_bytes ? g->o.block_bytes : (64ull << 20);
    G_TRY(hhgt_reader_open(in->path.c_str(), bb, g->o.n_threads, 6, &in->rd));
    in->is_bgzf = hhgt_reader_is_bgzf(in->rd) != 0;
    return true;
}

bool set_header(hhgt_ingest *g, Input *in, const uint8_t *p, size_t n, bool at_eof)
{
    size_t hb = 0;
    uint64_t S = 0;
    if (!parse_header_text(p, n, &hb, &S, at_eof)) {
        hhgt_set_error("%s: no #CHROM header line in the first block of the VCF", in->kind ? "<memory>" : in->path.c_str());
        fail(g, HHGT_ERR_MALFORMED, hhgt_last_error());
        return false;
    }
    in->header.assign(reinterpret_cast<const char *>(p), hb);
    in->header_lines = 0;
    for (size_t i = 0; i < hb; ++i) in->header_lines += p[i] == '\n';
    in->S_file = S;
    in->S = g->o.sites_only ? 0 : S;
    in->st.n_samples = in->S;
    return true;
}

// host reader (BGZF / gzip / plain file): pinned ring -> device text buffers
bool run_reader_input(hhgt_ingest *g, Input *in)
{
    if (!open_reader(g, in)) return false;
    hipEvent_t cev[2] = {nullptr, nullptr};
    G_HIP(hipEventCreateWithFlags(&cev[0], wait_event_flags()));
    G_HIP(hipEventCreateWithFlags(&cev[1], wait_event_flags()));
    int prev_tok = -1, k = 0;
    bool first = true, ok = true;
    for (;;) {
        const void *ptr = nullptr;
        uint64_t n = 0;
        int tok = -1, last = 0;
        const int rc = hhgt_reader_acquire(in->rd, &ptr, &n, &tok, &last);
        if (rc != HHGT_OK) {
            fail(g, rc, hhgt_last_error());
            ok = false;
            break;
        }
        if (n == 0) {
            if (first) {
                hhgt_set_error("%s: empty file (no VCF header)", in->path.c_str());
                fail(g, HHGT_ERR_MALFORMED, hhgt_last_error());
                ok = false;
                break;
            }
            // The reader may end a stream with an EMPTY last block (a gzip stream whose text fills a block exactly before
            // zlib reports the end).  Blocks were already sent without `last`, so an empty one with last = true still has to go
            // down the pipe: the harvest of THAT block pads and frames the open chunk column and sends INPUT_END — without it
            // up to vc - 1 variants' genotypes would silently be missing behind their variant-table rows.
            int ti;
            if (!take_text(g, &ti, 64)) {
                ok = false;
                break;
            }
            TextBuf &tb = g->text[(size_t)ti];
            if (hipEventRecord(tb.ready, g->s_copy) != hipSuccess) {
                fail(g, HHGT_ERR_HIP, "ingest: event record failed");
                ok = false;
                break;
            }
            tb.nbytes = 0;
            tb.in = in;
            tb.first = false;
            tb.last = true;
            if (!push_text(g, ti)) ok = false;
            break;
        }
        if (first && !set_header(g, in, static_cast<const uint8_t *>(ptr), (size_t)n, last != 0)) {
            ok = false;
            break;
        }
        int ti;
        if (!take_text(g, &ti, (size_t)n + 64)) {
            ok = false;
            break;
        }
        TextBuf &tb = g->text[(size_t)ti];
        // (a device-inflated input before this one may still have a carry copy out of this buffer queued on its own stream)
        if ((g->last_carry && hipStreamWaitEvent(g->s_copy, g->last_carry, 0) != hipSuccess) ||
            hipMemcpyAsync(tb.d, ptr, n, hipMemcpyHostToDevice, g->s_copy) != hipSuccess ||
            hipEventRecord(tb.ready, g->s_copy) != hipSuccess || hipEventRecord(cev[k & 1], g->s_copy) != hipSuccess) {
            fail(g, HHGT_ERR_HIP, "ingest: upload of a text block failed");
            ok = false;
            break;
        }
        tb.nbytes = n;
        tb.in = in;
        tb.first = first;
        tb.last = last != 0;
        first = false;
        if (!push_text(g, ti)) {
            ok = false;
            break;
        }
        // the previous pinned block goes back to the reader once its copy has left it
        if (prev_tok >= 0) {
            hipEventSynchronize(cev[(k + 1) & 1]);
            hhgt_reader_release(in->rd, prev_tok);
        }
        prev_tok = tok;
        ++k;
        if (last) break;
    }
    if (prev_tok >= 0) {
        hipEventSynchronize(cev[(k + 1) & 1]);
        hhgt_reader_release(in->rd, prev_tok);
    }
    hipEventDestroy(cev[0]);
    hipEventDestroy(cev[1]);
    uint64_t fb = 0, tbytes = 0;
    hhgt_reader_stats(in->rd, &fb, &tbytes);
    in->st.file_bytes = fb;
    hhgt_reader_close(in->rd);
    in->rd = nullptr;
    return ok;
}

// text already in host memory: cut at line ends, upload
bool run_memory_input(hhgt_ingest *g, Input *in)
{
    const uint64_t bb = g->o.block_bytes ? g->o.block_bytes : (64ull << 20);
    const uint8_t *p = in->mem;
    const uint64_t N = in->mem_bytes;
    if (N == 0) {
        fail(g, HHGT_ERR_MALFORMED, "<memory>: empty text (no VCF header)");
        return false;
    }
    if (!set_header(g, in, p, (size_t)(N < bb ? N : bb), N <= bb)) return false;
    in->st.file_bytes = N;
    uint64_t pos = 0;
    bool first = true;
    while (pos < N) {
        uint64_t n = N - pos < bb ? N - pos : bb;
        const bool last = pos + n >= N;
        if (!last) {
            const void *nl = memrchr(p + pos, '\n', (size_t)n);
            if (!nl) {
                fail(g, HHGT_ERR_IO, "a line is longer than the text block size");
                return false;
            }
            n = (uint64_t)((const uint8_t *)nl - (p + pos)) + 1;
        }
        int ti;
        if (!take_text(g, &ti, (size_t)n + 64)) return false;
        TextBuf &tb = g->text[(size_t)ti];
        if (g->last_carry) G_HIP(hipStreamWaitEvent(g->s_copy, g->last_carry, 0));   // (see run_reader_input)
        G_HIP(hipMemcpyAsync(tb.d, p + pos, n, hipMemcpyHostToDevice, g->s_copy));
        G_HIP(hipEventRecord(tb.ready, g->s_copy));
        tb.nbytes = n;
        tb.in = in;
        tb.first = first;
        tb.last = last;
        first = false;
        if (!push_text(g, ti)) return false;
        pos += n;
    }
    return true;
}

__global__ void k_count_bad_members(const uint32_t *__restrict__ st, uint64_t n, unsigned long long *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && st[i] != 0u);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

// BGZF file inflated on the device: the host walks the member headers and decides the block cuts, the compressed
// members cross PCIe, one wave per member writes the text (csrc/inflate.hip).
// Per block the source thread: pread()s the next stretch of the file straight into a pinned staging slot (no mapping:
// read() copies out of the page cache without faulting a page per 4 KiB), walks the member headers there until the
// block's text budget is used up, inflates the LAST member(s) on the host to learn where the last whole line ends,
// writes the member tables behind the compressed bytes and queues upload + carry copy + inflate kernels.  The first
// blocks of an input are small (the GPU starts after ~1 ms of host work), later ones grow to block_bytes (a launch
// wants >= 10 k members to fill the chip).
bool run_device_inflate_input(hhgt_ingest *g, Input *in)
{
    const uint64_t bb = g->o.block_bytes ? g->o.block_bytes : (512ull << 20);
    int fd = open(in->path.c_str(), O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        if (fd >= 0) close(fd);
        hhgt_set_error("cannot open %s", in->path.c_str());
        fail(g, HHGT_ERR_IO, hhgt_last_error());
        return false;
    }
    const uint64_t flen = (uint64_t)st.st_size;
    in->st.file_bytes = flen;
    in->is_bgzf = true;
    bool ok = true;
    trace("src:file_open", (long long)flen);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    std::vector<uint8_t> scratch(65536);
    uint64_t *c_off = nullptr;
    uint32_t *c_len = nullptr, *isz = nullptr, *crc = nullptr;
    auto host_inflate = [&](const uint8_t *base, size_t m, uint8_t *dst) -> bool {
        if (inflateReset(&zs) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(base + c_off[m]);
        zs.avail_in = c_len[m];
        zs.next_out = dst;
        zs.avail_out = isz[m];
        return inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
    };
    auto pread_all = [&](uint8_t *dst, size_t n, uint64_t off) -> bool {
        size_t got = 0;
        while (got < n) {
            const ssize_t k = pread(fd, dst + got, n - got, (off_t)(off + got));
            if (k <= 0) return false;
            got += (size_t)k;
        }
        return true;
    };
    uint64_t fpos = 0;          // file offset of the first member not yet in a block
    uint64_t carry = 0;         // bytes of the previous block behind its last newline
    int prev_ti = -1;
    uint64_t prev_cut = 0;
    bool first = true;
    double ratio = 24.0;        // text bytes per file byte, refined as blocks go by
    // text budget of the first block: small when nothing is in flight (the GPU starts after ~1 ms of host work instead
    // of ~5), full size when earlier inputs still keep the device busy (a small launch fills a fraction of the chip)
    const bool idle = g->free_text.size() == g->text.size() && g->q_text.size() == 0;
    uint64_t budget = idle ? (bb < (96ull << 20) ? bb : (96ull << 20)) : bb;
    uint64_t member_index = 0;
    if (flen == 0) {
        fail(g, HHGT_ERR_MALFORMED, "empty file (no VCF header)");
        ok = false;
    }
    while (ok && fpos < flen) {
        Staging &sg = g->stg[g->stg_next];
        g->stg_next = (g->stg_next + 1) % N_STG;
        if (sg.used) hipEventSynchronize(sg.done);   // the inflate that read this staging slot two blocks ago
        trace("src:staging_free");
        // stretch of the file to look at: what the budget should need, plus slack; at least one whole member
        uint64_t want = (uint64_t)((double)budget / ratio * 1.15) + (256u << 10);
        if (want > flen - fpos) want = flen - fpos;
        const size_t tab_room = (size_t)(want / 26 + 2) * 28 + 64;   // tables live behind the compressed bytes (a member is >= 26 bytes)
        if (sg.h.ensure((size_t)want + 64 + tab_room) != HHGT_OK) {
            fail(g, HHGT_ERR_HIP, hhgt_last_error());
            ok = false;
            break;
        }
        {
            // one thread copies ~5 GB/s out of the page cache; a 1 GiB text block needs ~40 MB of file
            const size_t piece = 2u << 20;
            const int nth = want > 4 * piece ? 4 : 1;
            std::atomic<size_t> next{0};
            std::atomic<bool> rd_ok{true};
            auto work = [&] {
                for (;;) {
                    const size_t o = next.fetch_add(piece);
                    if (o >= want) break;
                    const size_t n = want - o < piece ? (size_t)(want - o) : piece;
                    if (!pread_all(sg.h.p + o, n, fpos + o)) rd_ok.store(false);
                }
            };
            std::vector<std::thread> th;
            for (int i = 1; i < nth; ++i) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            if (!rd_ok.load()) {
                fail(g, HHGT_ERR_IO, "read failed");
                ok = false;
                break;
            }
        }
        trace("src:read_done", (long long)want);
        // member table of the stretch, cut at the text budget
        const size_t max_m = (size_t)(want / 26 + 2);
        g->mtab.ensure(max_m);
        c_off = g->mtab.c_off.get();
        c_len = g->mtab.c_len.get();
        isz = g->mtab.isz.get();
        crc = g->mtab.crc.get();
        uint64_t nm64 = 0, used = 0;
        const int rc = hhgt_bgzf_scan(sg.h.p, want, max_m, c_off, c_len, isz, crc, &nm64, &used);
        if (rc != HHGT_OK) {
            fail(g, rc, hhgt_last_error());
            ok = false;
            break;
        }
        if (nm64 == 0) {
            if (want == flen - fpos) fail(g, HHGT_ERR_MALFORMED, "bytes behind the last whole BGZF member");
            else fail(g, HHGT_ERR_IO, "a BGZF member larger than the staging stretch");
            ok = false;
            break;
        }
        size_t nm = 0;
        uint64_t total = 0, consumed = 0;
        while (nm < nm64 && carry + total + isz[nm] <= budget) {
            total += isz[nm];
            consumed = c_off[nm] + c_len[nm] + 8;   // payload + CRC32 + ISIZE
            ++nm;
        }
        if (nm == 0) {
            fail(g, HHGT_ERR_IO, "a line is longer than the text block: raise block_bytes");
            ok = false;
            break;
        }
        const bool last = fpos + consumed >= flen;
        trace("src:scan_done", (long long)nm);
        if (first) {
            // header: leading members inflated on the host until the '#' lines are complete
            std::vector<uint8_t> head;
            bool have = false;
            for (size_t m = 0; m < nm64 && head.size() < (256u << 20); ++m) {
                const size_t at = head.size();
                head.resize(at + isz[m]);
                if (isz[m] && !host_inflate(sg.h.p, m, head.data() + at)) {
                    fail(g, HHGT_ERR_IO, "inflate failed (header members)");
                    ok = false;
                    break;
                }
                size_t hb;
                uint64_t S;
                const bool eof = fpos + c_off[m] + c_len[m] + 8 >= flen;
                if (parse_header_text(head.data(), head.size(), &hb, &S, eof)) {
                    have = set_header(g, in, head.data(), head.size(), eof);
                    break;
                }
            }
            if (!ok) break;
            if (!have) {
                if (!g->failed.load()) {
                    hhgt_set_error("%s: no #CHROM header line", in->path.c_str());
                    fail(g, HHGT_ERR_MALFORMED, hhgt_last_error());
                }
                ok = false;
                break;
            }
        }
        if (first) trace("src:header_done");
        // bytes behind the last newline move to the next block: the last member(s) are inflated here to find it
        uint64_t tail = 0;
        if (!last) {
            bool found = false;
            for (size_t m = nm; m > 0 && !found; --m) {
                const size_t mm = m - 1;
                if (isz[mm] == 0) continue;
                if (!host_inflate(sg.h.p, mm, scratch.data())) {
                    fail(g, HHGT_ERR_IO, "inflate failed (block tail)");
                    ok = false;
                    break;
                }
                const void *nl = memrchr(scratch.data(), '\n', isz[mm]);
                if (nl) {
                    tail += isz[mm] - ((uint64_t)((const uint8_t *)nl - scratch.data()) + 1);
                    found = true;
                } else {
                    tail += isz[mm];
                }
            }
            if (!ok) break;
            if (!found) {
                fail(g, HHGT_ERR_IO, "a line is longer than the text block: raise block_bytes");
                ok = false;
                break;
            }
        }
        int ti;
        if (!take_text(g, &ti, (size_t)(carry + total) + 64)) {
            ok = false;
            break;
        }
        TextBuf &tb = g->text[(size_t)ti];
        trace("src:took_text", (long long)ti);
        // staging layout: [compressed bytes | comp_off u64 | out_off u64 | comp_len u32 | isize u32 | crc u32]
        const size_t comp_bytes = ((size_t)consumed + 3) / 4 * 4 + 4;
        const size_t o_coff = (comp_bytes + 7) & ~(size_t)7, o_ooff = o_coff + nm * 8, o_clen = o_ooff + nm * 8,
                     o_isz = o_clen + nm * 4, o_crc = o_isz + nm * 4, stg_bytes = o_crc + nm * 4;
        if (stg_bytes > sg.h.cap) {
            fail(g, HHGT_ERR_IO, "ingest: staging slot too small for the member tables");
            ok = false;
            break;
        }
        if (sg.d.ensure(stg_bytes) != HHGT_OK || tb.status.ensure(nm * 4) != HHGT_OK || tb.bad.ensure(8) != HHGT_OK) {
            fail(g, HHGT_ERR_HIP, hhgt_last_error());
            ok = false;
            break;
        }
        memset(sg.h.p + consumed, 0, comp_bytes - (size_t)consumed);   // the kernel reads whole dwords
        uint64_t *hc = reinterpret_cast<uint64_t *>(sg.h.p + o_coff), *ho = reinterpret_cast<uint64_t *>(sg.h.p + o_ooff);
        uint32_t *hl = reinterpret_cast<uint32_t *>(sg.h.p + o_clen), *hz = reinterpret_cast<uint32_t *>(sg.h.p + o_isz),
                 *hr = reinterpret_cast<uint32_t *>(sg.h.p + o_crc);
        uint64_t oo = carry;
        for (size_t i = 0; i < nm; ++i) {
            hc[i] = c_off[i];
            ho[i] = oo;
            hl[i] = c_len[i];
            hz[i] = isz[i];
            hr[i] = crc[i];
            oo += isz[i];
        }
        uint8_t *dd = sg.d.as<uint8_t>();
        hipStream_t si = (g->inf_blocks++ & 1) ? g->s_inf2 : g->s_inf;
        hipError_t e = hipMemcpyAsync(dd, sg.h.p, stg_bytes, hipMemcpyHostToDevice, si);
        // this block may write a text buffer an earlier carry copy still reads from (they run on their own stream)
        if (e == hipSuccess && g->last_carry) e = hipStreamWaitEvent(si, g->last_carry, 0);
        if (e == hipSuccess && carry) {
            TextBuf &pb = g->text[(size_t)prev_ti];
            e = hipStreamWaitEvent(g->s_carry, pb.ready, 0);   // the previous block's text is complete
            if (e == hipSuccess) e = hipMemcpyAsync(tb.d, pb.d + prev_cut, carry, hipMemcpyDeviceToDevice, g->s_carry);
            if (e == hipSuccess) e = hipEventRecord(tb.carry_done, g->s_carry);
            g->last_carry = tb.carry_done;
            // free_text is a FIFO of four, so a block never lands in the buffer of the one before it; if it ever does, the
            // copy has to read the tail before the inflate overwrites it
            if (e == hipSuccess && ti == prev_ti) e = hipStreamWaitEvent(si, tb.carry_done, 0);
        }
        if (e == hipSuccess) e = hipMemsetAsync(tb.bad.p, 0, 8, si);
        if (e != hipSuccess) {
            fail(g, HHGT_ERR_HIP, "ingest: upload of compressed members failed");
            ok = false;
            break;
        }
        const int rc2 = launch_inflate(dd, comp_bytes, reinterpret_cast<const uint64_t *>(dd + o_coff),
                                       reinterpret_cast<const uint32_t *>(dd + o_clen), reinterpret_cast<const uint64_t *>(dd + o_ooff),
                                       reinterpret_cast<const uint32_t *>(dd + o_isz), nm, tb.d, tb.cap, tb.status.as<uint32_t>(),
                                       reinterpret_cast<const uint32_t *>(dd + o_crc), g->crc_x2n.as<uint32_t>(), si);
        if (rc2 != HHGT_OK) {
            fail(g, rc2, hhgt_last_error());
            ok = false;
            break;
        }
        hipLaunchKernelGGL(k_count_bad_members, dim3((uint32_t)((nm + 255) / 256)), dim3(256), 0, si, tb.status.as<uint32_t>(),
                           (uint64_t)nm, tb.bad.as<unsigned long long>());
        *tb.h_bad = 0;
        e = hipMemcpyAsync(tb.h_bad, tb.bad.p, 8, hipMemcpyDeviceToHost, si);
        if (e == hipSuccess && carry) e = hipStreamWaitEvent(si, tb.carry_done, 0);
        if (e == hipSuccess) e = hipEventRecord(tb.ready, si);
        if (e == hipSuccess) e = hipEventRecord(sg.done, si);
        if (e != hipSuccess) {
            fail(g, HHGT_ERR_HIP, "ingest: device inflate launch failed");
            ok = false;
            break;
        }
        sg.used = true;
        tb.nbytes = carry + total - tail;
        tb.in = in;
        tb.first = first;
        tb.last = last;
        tb.n_members = nm;
        tb.first_member = member_index;
        member_index += nm;
        first = false;
        prev_ti = ti;
        prev_cut = tb.nbytes;
        carry = tail;
        fpos += consumed;
        if (total && consumed) ratio = 0.5 * ratio + 0.5 * ((double)total / (double)consumed);
        budget = budget * 4 < bb ? budget * 4 : bb;
        if (!push_text(g, ti)) ok = false;
    }
    inflateEnd(&zs);
    close(fd);
    return ok;
}

void source_main(hhgt_ingest *g)
{
    hipSetDevice(g->device);
    for (;;) {
        Input *in = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->in_mu);
            g->in_cv.wait(lk, [&] { return g->failed.load() || g->finished || g->next_input < g->inputs.size(); });
            if (g->failed.load()) break;
            if (g->next_input >= g->inputs.size()) break;   // finished and drained
            in = g->inputs[g->next_input++].get();
        }
        if (in->kind == 0) in->dev_inflate = wants_device_inflate(g->o.device_inflate, in->path.c_str());
        in->st.device_inflate = in->dev_inflate ? 1 : 0;
        // open the readers of the next file inputs now: their inflate runs while this input is uploaded and encoded
        if (!in->dev_inflate) {
            std::vector<Input *> ahead;
            {
                std::lock_guard<std::mutex> lk(g->in_mu);
                const int A = g->o.files_ahead > 0 ? g->o.files_ahead : 1;
                for (size_t i = g->next_input; i < g->inputs.size() && (int)ahead.size() < A; ++i)
                    if (g->inputs[i]->kind == 0 && !g->inputs[i]->rd) ahead.push_back(g->inputs[i].get());
            }
            for (Input *a : ahead)
                if (!wants_device_inflate(g->o.device_inflate, a->path.c_str()) && !open_reader(g, a)) break;
        }
        bool ok = in->kind == 1 ? run_memory_input(g, in) : (in->dev_inflate ? run_device_inflate_input(g, in) : run_reader_input(g, in));
        in->st.is_bgzf = in->is_bgzf ? 1 : 0;
        if (!ok || g->failed.load()) break;
    }
    // readers opened ahead but never run (error paths)
    {
        std::lock_guard<std::mutex> lk(g->in_mu);
        for (auto &in : g->inputs)
            if (in->rd) {
                hhgt_reader_close(in->rd);
                in->rd = nullptr;
            }
    }
    g->q_text.push(-1);
}

// ---------------------------------------------------------------------------------------------------------------
// driver thread
// ---------------------------------------------------------------------------------------------------------------
// geometry of an input's ring + every buffer whose size follows from the sample count: the state's own (ring of chunk columns
// as planes / int8, variant tables, cursor) and the batch slots.  at_open: called from hhgt_ingest_open with the caller's
// expect_samples, when every slot still sits in its pool — so that the first input finds its buffers made (and pinned).
static bool size_input_state(hhgt_ingest *g, hhgt_ingest::InState *X, uint64_t S, uint64_t S_file, uint64_t block_bytes, bool at_open)
{
    const int32_t sc = g->o.sc, vc = g->o.vc;
    // a kept line holds S sample columns of at least two bytes behind nine fixed columns
    // (file inputs: the reader never hands out blocks below 1 MiB and grows a text buffer to what a block needs)
    if (block_bytes < (1ull << 20)) block_bytes = 1ull << 20;
    X->kept_per_block = block_bytes / (2 * S_file + 16) + 2;
    const uint64_t W = X->kept_per_block / (uint64_t)vc + 2;   // chunk columns one block can touch
    X->ring_cols = 2 * W + 6;
    memset(&X->lay, 0, sizeof(X->lay));
    X->lay.n_samples = (int32_t)S;
    X->lay.sc = sc;
    X->lay.vc = vc;
    X->lay.ring = (int32_t)X->ring_cols;
    X->lay.v_capacity = X->ring_cols * (uint64_t)vc;
    X->n_sc = S ? (S + (uint64_t)sc - 1) / (uint64_t)sc : 0;
    X->chunk_nbytes = (uint64_t)sc * (uint64_t)vc * 2;
    X->col_bytes = X->n_sc * X->chunk_nbytes;
    const uint64_t gbytes = hhgt_layout_bytes(&X->lay);
    G_TRY(X->G.ensure((size_t)(gbytes ? gbytes : 16)));
    // the compressor is this engine's only consumer of the matrix: where the chunk geometry allows (typesize 2, 8 KiB Blosc
    // blocks, whole blocks per chunk row) the encoder hands it over as bit planes — a quarter of the bytes each way
    static const bool planes_env = !(getenv("HHGT_INGEST_PLANES") && atoi(getenv("HHGT_INGEST_PLANES")) == 0);
    X->planes = planes_env && S > 0 && g->o.typesize == 2 && g->o.blocksize == 8192 && vc % 4096 == 0 && hhgt_planes_bytes(&X->lay) != 0;
    if (X->planes) G_TRY(X->P.ensure((size_t)hhgt_planes_bytes(&X->lay)));
    G_TRY(X->t_start.ensure((size_t)X->lay.v_capacity * 4));
    G_TRY(X->t_ref.ensure((size_t)X->lay.v_capacity));
    G_TRY(X->t_alt.ensure((size_t)X->lay.v_capacity));
    G_TRY(X->cursor.ensure(8));
    // batch buffers.  They may still be in use by batches of the previous input that are on their way out, so when
    // one has to grow (this input has more samples, or is the first) every slot is collected first — the shipper and
    // the consumer give them back as they go — and returned to the pools afterwards.
    const uint64_t max_cols = W + 2;
    const size_t need_d = (size_t)(max_cols * X->n_sc * (X->chunk_nbytes + 32) + 64), need_off = (size_t)((max_cols * X->n_sc + 1) * 8);
    bool grow = false;
    for (auto &d : g->dst) grow = grow || d.d.cap < need_d || d.off.cap < need_off || d.h_off.cap < need_off;
    for (auto &v : g->var) grow = grow || v.start.cap < (size_t)X->kept_per_block * 4 + 16 || v.ref.cap < (size_t)X->kept_per_block + 4;
    if (grow) {
        int tmp;
        if (!at_open) {
            for (int i = 0; i < N_DST; ++i)
                if (!g->free_dst.pop(tmp)) return false;
            for (int i = 0; i < N_VAR; ++i)
                if (!g->free_var.pop(tmp)) return false;
        }
        for (auto &d : g->dst) {
            G_TRY(d.d.ensure(need_d));
            G_TRY(d.off.ensure(need_off));
            G_TRY(d.h_off.ensure(need_off));
        }
        for (auto &v : g->var) {
            G_TRY(v.start.ensure((size_t)X->kept_per_block * 4 + 16));   // k_tables_to_host writes whole groups of four
            G_TRY(v.ref.ensure((size_t)X->kept_per_block + 4));
            G_TRY(v.alt.ensure((size_t)X->kept_per_block + 4));
        }
        if (!at_open) {
            for (int i = 0; i < N_DST; ++i) g->free_dst.push(i);
            for (int i = 0; i < N_VAR; ++i) g->free_var.push(i);
        }
    }
    if (at_open) {
        // the shipper's pinned copies of the framed chunks: a batch is at most need_d bytes and about a fifth of that on
        // genotype planes — a slot that turns out too small still grows where it is used
        for (auto &o : g->out) G_TRY(o.h.ensure(need_d / 4 + 4096));
    }
    return true;
}

bool begin_input(hhgt_ingest *g, Input *in, uint64_t block_bytes)
{
    hhgt_ingest::InState *X = &g->ist[g->n_begun++ & 1u];
    in->state = X;
    const uint64_t S = in->S;
    if (!size_input_state(g, X, S, in->S_file, block_bytes, false)) return false;
    const uint64_t gbytes = hhgt_layout_bytes(&X->lay);
    const int32_t sc = g->o.sc;
    // sample padding rows (S .. round_up(S, sc)) are never written by the encoder: zeroed once per input for every
    // ring column (everything else of a column is overwritten, or zeroed by the tail padding, before it is framed)
    if (gbytes && S % (uint64_t)sc) {
        if (X->planes) G_TRY(hhgt_pad_tail_planes(g->ctx, &X->lay, X->lay.v_capacity, 0, X->ring_cols, X->P.p, g->s_main));
        else G_TRY(hhgt_pad_tail(g->ctx, &X->lay, X->lay.v_capacity, 0, X->ring_cols, X->G.p, g->s_main));
    }
    G_HIP(hipMemsetAsync(X->cursor.p, 0, 8, g->s_main));
    X->done_cols = 0;
    X->host_cursor = 0;
    in->t_first = now_s();
    in->header_sent = false;   // announced by the first harvest of this input: behind the previous input's last events
    return true;
}

// rows [a, a + n) of the three variant tables (rings of `cap` entries) -> pinned host memory, written by the kernel itself over
// the link.  One launch instead of three to six hipMemcpyAsync calls — and the runtime's copy path is kept out of the driver
// thread: HHGT_INGEST_DEBUG=2 showed the first such copy above ~200 KB blocking the calling thread for as long as the encode
// queued behind it on the stream ran (7 ms of a 60 ms first pass).  Thread i moves variants 4i .. 4i+3.
__global__ void __launch_bounds__(256) k_tables_to_host(const uint32_t *__restrict__ t_start, const uint8_t *__restrict__ t_ref,
                                                        const uint8_t *__restrict__ t_alt, uint64_t a, uint64_t n, uint64_t cap,
                                                        uint32_t *__restrict__ h_start, uint32_t *__restrict__ h_ref,
                                                        uint32_t *__restrict__ h_alt)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x, v0 = i * 4;
    if (v0 >= n) return;
    uint32_t st[4] = {0, 0, 0, 0}, r = 0, al = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (v0 + k < n) {
            const uint64_t s = (a + v0 + k) % cap;
            st[k] = t_start[s];
            r |= (uint32_t)t_ref[s] << (8 * k);
            al |= (uint32_t)t_alt[s] << (8 * k);
        }
    *reinterpret_cast<uint4 *>(h_start + v0) = make_uint4(st[0], st[1], st[2], st[3]);   // slots hold kept_per_block + 4 entries
    h_ref[i] = r;
    h_alt[i] = al;
}

bool get_event(hhgt_ingest *g, hipEvent_t *ev)
{
    return g->free_ev.pop(*ev);
}

// completed columns [c0, c1) -> compress batches (one per contiguous run of ring slots)
bool queue_columns(hhgt_ingest *g, Input *in, uint64_t c0, uint64_t c1)
{
    hhgt_ingest::InState *X = static_cast<hhgt_ingest::InState *>(in->state);
    while (c0 < c1) {
        const uint64_t slot = c0 % X->ring_cols;
        const uint64_t n = (c1 - c0) < (X->ring_cols - slot) ? (c1 - c0) : (X->ring_cols - slot);
        Batch b;
        b.kind = B_COLUMNS;
        b.in = in;
        b.first_col = c0;
        b.n_cols = n;
        b.n_chunks = n * X->n_sc;
        b.raw_bytes = n * X->col_bytes;
        if (!g->free_dst.pop(b.dst_slot) || !get_event(g, &b.ev)) return false;
        trace("drv:dst_slot", (long long)n);
        DstSlot &d = g->dst[(size_t)b.dst_slot];
        const int bs = g->o.blocksize;
        if (X->planes)
            G_TRY(hhgt_compress_planes(g->ctx, &X->lay, X->P.p, X->G.p, (uint32_t)slot, (uint32_t)n, g->o.format, d.d.p, d.d.cap,
                                       d.off.as<uint64_t>(), nullptr, g->s_main));
        else
            G_TRY(hhgt_compress_chunks(g->ctx, X->G.as<uint8_t>() + slot * X->col_bytes, b.n_chunks, X->chunk_nbytes, g->o.typesize, bs,
                                       g->o.format, d.d.p, d.d.cap, d.off.as<uint64_t>(), nullptr, g->s_main));
        trace("drv:compress_queued", (long long)b.n_chunks);
        G_HIP(hipMemcpyAsync(d.h_off.p, d.off.p, (size_t)((b.n_chunks + 1) * 8), hipMemcpyDeviceToHost, g->s_main));
        G_HIP(hipEventRecord(b.ev, g->s_main));
        trace("drv:off_copy_queued");
        g->q_ship.push(b);
        c0 += n;
    }
    return true;
}

// look at the result record of an encoded block (its event has been waited for)
bool harvest_body(hhgt_ingest *g, hhgt_ingest::Res &r);
bool harvest(hhgt_ingest *g, hhgt_ingest::Res &r)
{
    const double t0 = now_s();
    G_HIP(hipEventSynchronize(r.ev));
    const double t1 = now_s();
    trace("drv:encode_done", r.text_idx);
    const bool ok = harvest_body(g, r);
    trace("drv:harvested");
    g->tm.drv_harvest_wait += t1 - t0;
    g->tm.drv_harvest += now_s() - t1;
    return ok;
}

bool harvest_body(hhgt_ingest *g, hhgt_ingest::Res &r)
{
    TextBuf &tb = g->text[(size_t)r.text_idx];
    Input *in = tb.in;
    hhgt_ingest::InState *X = r.st;
    if (!in->header_sent) {
        Batch h;
        h.kind = B_HEADER;
        h.in = in;
        g->q_ship.push(h);
        in->header_sent = true;
    }
    const hhgt_encode_result rec = *r.rec;
    if (in->dev_inflate && *tb.h_bad) {
        // which member, and why: only now is the per-member status worth copying
        std::vector<uint32_t> st((size_t)tb.n_members);
        hipMemcpy(st.data(), tb.status.p, st.size() * 4, hipMemcpyDeviceToHost);
        size_t k = 0;
        while (k < st.size() && !st[k]) ++k;
        const uint32_t code = k < st.size() ? st[k] : 0;
        if (code == 9) hhgt_set_error("BGZF member %llu: CRC-32 of the inflated text differs from the trailer", (unsigned long long)(tb.first_member + k));
        else hhgt_set_error("BGZF member %llu: DEFLATE stream is corrupt (status %u)", (unsigned long long)(tb.first_member + k), code);
        fail(g, HHGT_ERR_IO, hhgt_last_error());
        return false;
    }
    if (rec.n_lines_over && !rec.err_density) {
        hhgt_set_error("Error parsing VCF file: %llu more lines than records of %llu samples fit in the block (blank or cut-off lines)",
                       (unsigned long long)rec.n_lines_over, (unsigned long long)in->S_file);
        fail(g, HHGT_ERR_MALFORMED, hhgt_last_error());
        return false;
    }
    G_TRY(hhgt_encode_result_status(&rec));
    if (rec.stats.n_chrom_runs > HHGT_RESULT_RUNS) {
        fail(g, HHGT_ERR_CAPACITY, "more than 16 CHROM runs in one text block: the input is not sorted by contig");
        return false;
    }
    in->st.n_lines += rec.stats.n_lines;
    in->st.n_records += rec.stats.n_records;
    in->st.n_drop_region += rec.stats.n_drop_region;
    in->st.n_drop_filter += rec.stats.n_drop_filter;
    in->st.n_haploid_padded += rec.stats.n_haploid_padded;
    in->st.n_general_lines += rec.stats.n_general_lines;
    in->st.text_bytes += tb.nbytes;
    in->st.n_blocks += 1;
    const uint64_t a = rec.cursor_before, b = rec.cursor_after;
    const bool last = tb.last;
    if (b - a > X->kept_per_block) {   // the ring and the variant slots were sized from this bound: never trust it silently
        hhgt_set_error("ingest: a text block kept %llu records, the engine's buffers were sized for %llu (block of %llu bytes)",
                       (unsigned long long)(b - a), (unsigned long long)X->kept_per_block, (unsigned long long)tb.nbytes);
        fail(g, HHGT_ERR_CAPACITY, hhgt_last_error());
        return false;
    }
    g->free_text.push(r.text_idx);   // the text has been consumed: the source may overwrite the buffer
    r.text_idx = -1;
    X->host_cursor = b;
    in->st.n_kept = b;
    if (b > a || rec.stats.n_chrom_runs) {
        Batch v;
        v.kind = B_VARIANTS;
        v.in = in;
        v.first_variant = a;
        v.n_variants = b - a;
        if (!g->free_var.pop(v.var_slot) || !get_event(g, &v.ev)) return false;
        trace("drv:var_slot");
        VarSlot &vs = g->var[(size_t)v.var_slot];
        const uint64_t cap = X->lay.v_capacity;
        if (b > a) {
            const uint64_t nt = (b - a + 3) / 4;
            hipLaunchKernelGGL(k_tables_to_host, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, g->s_main, X->t_start.as<uint32_t>(),
                               X->t_ref.as<uint8_t>(), X->t_alt.as<uint8_t>(), a, b - a, cap, reinterpret_cast<uint32_t *>(vs.start.p),
                               reinterpret_cast<uint32_t *>(vs.ref.p), reinterpret_cast<uint32_t *>(vs.alt.p));
            G_HIP(hipGetLastError());
        }
        trace("drv:tables_queued", (long long)(b - a));
        for (uint64_t i = 0; i < rec.stats.n_chrom_runs; ++i) {
            const std::string name(rec.run_names[i], strnlen(rec.run_names[i], 31));
            if (name == in->last_run) continue;   // the block continues the previous block's contig
            in->last_run = name;
            vs.run_first[v.n_runs] = a + rec.run_first[i];
            memset(vs.run_names[v.n_runs], 0, 32);
            memcpy(vs.run_names[v.n_runs], name.data(), name.size());
            ++v.n_runs;
        }
        G_HIP(hipEventRecord(v.ev, g->s_main));
        g->q_ship.push(v);
        trace("drv:var_queued");
    }
    if (in->S) {
        const uint64_t done = b / (uint64_t)g->o.vc;
        if (done > X->done_cols) {
            if (!queue_columns(g, in, X->done_cols, done)) return false;
            X->done_cols = done;
        }
        if (last && b % (uint64_t)g->o.vc) {
            // the open column: zero behind the cursor, frame it
            if (X->planes) G_TRY(hhgt_pad_tail_planes_cursor(g->ctx, &X->lay, X->cursor.as<uint64_t>(), X->P.p, g->s_main));
            else G_TRY(hhgt_pad_tail_cursor(g->ctx, &X->lay, X->cursor.as<uint64_t>(), X->G.p, g->s_main));
            if (!queue_columns(g, in, X->done_cols, X->done_cols + 1)) return false;
            X->done_cols += 1;
        }
    }
    if (last) {
        Batch e;
        e.kind = B_INPUT_END;
        e.in = in;
        g->q_ship.push(e);
    }
    return true;
}

void driver_main(hhgt_ingest *g)
{
    hipSetDevice(g->device);
    std::deque<int> pending;   // result slots in flight, oldest first
    int next_res = 0;
    for (;;) {
        int ti;
        const double tw = now_s();
        const bool got = g->q_text.pop(ti);
        g->tm.drv_wait_text += now_s() - tw;
        if (!got) break;
        if (ti < 0) break;   // end of inputs
        if (g->tm.t_first_text == 0) g->tm.t_first_text = now_s();
        TextBuf &tb = g->text[(size_t)ti];
        Input *in = tb.in;
        auto drv = [&]() -> bool {
            const double tb0 = now_s();
            if (tb.first && !begin_input(g, in, tb.cap)) return false;
            g->tm.drv_begin += now_s() - tb0;
            if ((int)pending.size() >= N_RES - 1) {
                if (!harvest(g, g->res[pending.front()])) return false;
                pending.pop_front();
            }
            hhgt_ingest::Res &r = g->res[next_res];
            hhgt_ingest::InState *X = static_cast<hhgt_ingest::InState *>(in->state);
            r.st = X;
            const double tl0 = now_s();
            trace("drv:encode_launch", ti, (long long)tb.nbytes);
            G_HIP(hipStreamWaitEvent(g->s_main, tb.ready, 0));
            // Bound on the block's line count (sizes the workspaces and every grid behind the index): a record of a
            // file with S sample columns has at least 2 S + 17 bytes, so apart from the header lines of the first
            // block more lines than that can only be blank or cut-off lines — which are a parse error anyway
            // (reported as such by harvest).  The unconditional bound, 1024 lines per 16 KiB region, would size the
            // grids for lines of 16 bytes: 10 M empty workgroups per 1 GiB block.
            const uint64_t n_regions = (tb.nbytes + 1 + INDEX_REGION - 1) / INDEX_REGION;
            uint64_t max_lines = tb.nbytes / (2 * in->S_file + 17) + (tb.first ? in->header_lines : 0) + 64;
            if (max_lines > n_regions * INDEX_CAP) max_lines = n_regions * INDEX_CAP;
            if (X->planes)
                G_TRY(hhgt_encode_text_planes_async(g->ctx, tb.d, tb.nbytes, in->region.c_str(), &X->lay, X->cursor.as<uint64_t>(),
                                                    (uint32_t)(max_lines > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : max_lines), X->P.p, X->G.p,
                                                    X->t_start.as<uint32_t>(), nullptr, X->t_ref.as<uint8_t>(), X->t_alt.as<uint8_t>(),
                                                    r.rec, g->s_main));
            else
                G_TRY(hhgt_encode_text_async(g->ctx, tb.d, tb.nbytes, in->region.c_str(), &X->lay, X->cursor.as<uint64_t>(),
                                             (uint32_t)(max_lines > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : max_lines), X->G.p,
                                             X->t_start.as<uint32_t>(), nullptr, X->t_ref.as<uint8_t>(), X->t_alt.as<uint8_t>(), r.rec,
                                             g->s_main));
            G_HIP(hipEventRecord(r.ev, g->s_main));
            r.text_idx = ti;
            pending.push_back(next_res);
            next_res = (next_res + 1) % N_RES;
            g->tm.drv_launch += now_s() - tl0;
            // one block behind: the GPU has the encode above queued while the host looks at the previous result.  The
            // last block of an input is only drained at once when nothing else is waiting to be queued.
            while (pending.size() > 1 || (tb.last && !pending.empty() && g->q_text.size() == 0)) {
                if (!harvest(g, g->res[pending.front()])) return false;
                pending.pop_front();
            }
            return true;
        };
        if (!drv()) break;
    }
    while (!g->failed.load() && !pending.empty()) {
        if (!harvest(g, g->res[pending.front()])) break;
        pending.pop_front();
    }
    if (!g->failed.load()) {
        Batch e;
        e.kind = B_END;
        g->q_ship.push(e);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// shipper thread: sizes -> device-to-host copy of the framed bytes -> out queue
// ---------------------------------------------------------------------------------------------------------------
void ship_main(hhgt_ingest *g)
{
    hipSetDevice(g->device);
    hipEvent_t cev = nullptr;
    hipEventCreateWithFlags(&cev, wait_event_flags());
    for (;;) {
        Batch b;
        if (!g->q_ship.pop(b)) break;
        const double ts0 = now_s();
        if (b.kind == B_VARIANTS) {
            if (hipEventSynchronize(b.ev) != hipSuccess) {
                fail(g, HHGT_ERR_HIP, "ingest: variant table copy failed");
                break;
            }
            g->free_ev.push(b.ev);
            b.ev = nullptr;
        } else if (b.kind == B_COLUMNS) {
            if (hipEventSynchronize(b.ev) != hipSuccess) {
                fail(g, HHGT_ERR_HIP, "ingest: compress failed");
                break;
            }
            g->free_ev.push(b.ev);
            b.ev = nullptr;
            g->tm.ship_wait_ev += now_s() - ts0;
            trace("ship:compress_done", (long long)b.n_cols);
            DstSlot &d = g->dst[(size_t)b.dst_slot];
            const uint64_t *off = reinterpret_cast<const uint64_t *>(d.h_off.p);
            b.framed_bytes = off[b.n_chunks];
            const double tso = now_s();
            if (!g->free_out.pop(b.out_slot)) break;
            g->tm.ship_wait_out += now_s() - tso;
            const double tsc = now_s();
            OutSlot &o = g->out[(size_t)b.out_slot];
            if (o.h.ensure((size_t)b.framed_bytes + 64) != HHGT_OK) {
                fail(g, HHGT_ERR_HIP, hhgt_last_error());
                break;
            }
            o.off.assign(off, off + b.n_chunks + 1);
            if (hipMemcpyAsync(o.h.p, d.d.p, (size_t)b.framed_bytes, hipMemcpyDeviceToHost, g->s_out) != hipSuccess ||
                hipEventRecord(cev, g->s_out) != hipSuccess || hipEventSynchronize(cev) != hipSuccess) {
                fail(g, HHGT_ERR_HIP, "ingest: copy of the framed chunks failed");
                break;
            }
            g->tm.ship_copy += now_s() - tsc;
            trace("ship:copied", (long long)b.framed_bytes, (long long)b.n_cols);
            g->free_dst.push(b.dst_slot);
            b.dst_slot = -1;
            b.in->st.raw_bytes += b.raw_bytes;
            b.in->st.compressed_bytes += b.framed_bytes;
        } else if (b.kind == B_INPUT_END) {
            b.in->st.seconds = now_s() - b.in->t_first;
            static const bool dbg = getenv("HHGT_INGEST_DEBUG") != nullptr;
            if (dbg) {   // per input, then reset (the threads' counters are only read here: a development aid)
                auto &t = g->tm;
                fprintf(stderr,
                        "[hhgt ingest] input %d: %.1f ms | source: wait for a text buffer %.1f | driver: wait for text %.1f, begin_input %.1f, "
                        "encode launch %.1f, harvest wait %.1f, harvest work %.1f | shipper: wait for compress %.1f, wait for an out slot %.1f, "
                        "copy %.1f (ms)\n",
                        b.in->index, b.in->st.seconds * 1e3, t.src_wait_text * 1e3, t.drv_wait_text * 1e3, t.drv_begin * 1e3, t.drv_launch * 1e3,
                        t.drv_harvest_wait * 1e3, t.drv_harvest * 1e3, t.ship_wait_ev * 1e3, t.ship_wait_out * 1e3, t.ship_copy * 1e3);
                t = hhgt_ingest::Times();
            }
        }
        const int kind = b.kind;
        if (kind == B_END) g->tm.t_end = now_s();
        g->q_out.push(b);
        if (kind == B_END) break;
    }
    if (cev) hipEventDestroy(cev);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------------------------------
extern "C" int hhgt_ingest_open(hhgt_ctx *ctx, const hhgt_ingest_opts *opts, hhgt_ingest **out)
{
    if (!ctx || !out) return HHGT_ERR_ARG;
    *out = nullptr;
    hhgt_ingest *g = new hhgt_ingest();
    g->ctx = ctx;
    g->device = ctx->device;
    if (opts) g->o = *opts;
    else memset(&g->o, 0, sizeof(g->o));
    if (g->o.sc <= 0) g->o.sc = 64;
    if (g->o.vc <= 0) g->o.vc = 8192;
    if (g->o.typesize <= 0) g->o.typesize = 2;
    if (g->o.blocksize <= 0) g->o.blocksize = g->o.vc * 2 < 8192 ? g->o.vc * 2 : 8192;
    if (g->o.format != HHGT_BLOSC1 && g->o.format != HHGT_BLOSC2) g->o.format = HHGT_BLOSC2;
    if ((g->o.sc & (g->o.sc - 1)) || g->o.vc % TILE_V) {
        hhgt_set_error("ingest: sc must be a power of two and vc a multiple of %d", TILE_V);
        delete g;
        return HHGT_ERR_ARG;
    }
    int rc = HHGT_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == HHGT_OK) {
            hhgt_set_error("ingest: %s failed: %s", what, hipGetErrorString(e));
            rc = HHGT_ERR_HIP;
        }
    };
    hip(hipSetDevice(g->device), "hipSetDevice");
    hip(hipStreamCreateWithFlags(&g->s_main, hipStreamNonBlocking), "stream");
    // the copy streams get the highest priority: on this platform pinned copies can run as blit kernels, and a blit
    // queued behind thousands of resident LZ4 / inflate waves crawled at 4-13 GB/s (a 50 MB batch took up to 12 ms,
    // the driver ran out of batch slots and the GPU idled)
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hip(hipStreamCreateWithPriority(&g->s_copy, hipStreamNonBlocking, prio_hi), "stream");
    // the inflate streams at the lowest priority (HHGT_INGEST_INF_PRIO=0: the default one): one wave per member lives for the
    // whole launch, so a freed wave slot should go to the encode / compress kernels of the block before, not to the next inflate
    static const bool inf_low = !(getenv("HHGT_INGEST_INF_PRIO") && atoi(getenv("HHGT_INGEST_INF_PRIO")) == 0);
    hip(hipStreamCreateWithPriority(&g->s_inf, hipStreamNonBlocking, inf_low ? prio_lo : 0), "stream");
    hip(hipStreamCreateWithPriority(&g->s_inf2, hipStreamNonBlocking, inf_low ? prio_lo : 0), "stream");
    hip(hipStreamCreateWithFlags(&g->s_carry, hipStreamNonBlocking), "stream");
    hip(hipStreamCreateWithPriority(&g->s_out, hipStreamNonBlocking, prio_hi), "stream");
    const bool dev = g->o.device_inflate != 0;
    const uint64_t bb_host = g->o.block_bytes ? g->o.block_bytes : (64ull << 20);
    const uint64_t bb_dev = g->o.block_bytes ? g->o.block_bytes : (512ull << 20);
    const uint64_t bb = dev ? (bb_dev > bb_host ? bb_dev : bb_host) : bb_host;
    g->text.resize(dev ? N_TEXT_DEV : N_TEXT_HOST);
    for (size_t i = 0; i < g->text.size() && rc == HHGT_OK; ++i) {
        TextBuf &tb = g->text[i];
        hip(hipMalloc(reinterpret_cast<void **>(&tb.d), (size_t)bb + 256), "hipMalloc(text block)");
        tb.cap = rc == HHGT_OK ? (size_t)bb + 256 : 0;
        hip(hipEventCreateWithFlags(&tb.ready, hipEventDisableTiming), "event");
        hip(hipEventCreateWithFlags(&tb.carry_done, hipEventDisableTiming), "event");
        hip(hipHostMalloc(reinterpret_cast<void **>(&tb.h_bad), 8, hipHostMallocDefault), "hipHostMalloc");
        if (rc == HHGT_OK) g->free_text.push((int)i);
    }
    for (auto &s : g->stg) hip(hipEventCreateWithFlags(&s.done, wait_event_flags()), "event");
    for (int i = 0; i < N_RES && rc == HHGT_OK; ++i) {
        hip(hipHostMalloc(reinterpret_cast<void **>(&g->res[i].rec), sizeof(hhgt_encode_result), hipHostMallocDefault), "hipHostMalloc");
        hip(hipEventCreateWithFlags(&g->res[i].ev, wait_event_flags()), "event");
    }
    for (int i = 0; i < N_VAR; ++i) g->free_var.push(i);
    for (int i = 0; i < N_DST; ++i) g->free_dst.push(i);
    for (int i = 0; i < N_OUT; ++i) g->free_out.push(i);
    for (int i = 0; i < N_VAR + N_DST + 4 && rc == HHGT_OK; ++i) {
        hipEvent_t e = nullptr;
        hip(hipEventCreateWithFlags(&e, wait_event_flags()), "event");
        if (e) {
            g->batch_events.push_back(e);
            g->free_ev.push(e);
        }
    }
    if (rc == HHGT_OK && dev) {
        uint32_t t[32];
        crc32_x2n_table(t);
        rc = g->crc_x2n.ensure(sizeof(t));
        if (rc == HHGT_OK) hip(hipMemcpy(g->crc_x2n.p, t, sizeof(t), hipMemcpyHostToDevice), "hipMemcpy");
    }
    if (rc != HHGT_OK) {
        hhgt_ingest_close(g);
        return rc;
    }
    if (g->o.expect_samples > 0) {
        // the caller knows the cohort's width: everything whose size follows from it is made (and pinned) now instead of
        // inside the first input — both ring states, the batch slots, the shipper's pinned copies, and for the device
        // inflater the pinned staging of a block's compressed members
        const uint64_t S = g->o.sites_only ? 0ull : (uint64_t)g->o.expect_samples;
        bool ok = size_input_state(g, &g->ist[0], S, (uint64_t)g->o.expect_samples, bb + 256, true) &&
                  size_input_state(g, &g->ist[1], S, (uint64_t)g->o.expect_samples, bb + 256, true);
        if (ok) {   // the context's own workspaces for a block / a batch of that size
            const hhgt_ingest::InState &X0 = g->ist[0];
            const uint64_t W = X0.kept_per_block / (uint64_t)g->o.vc + 2;
            ok = hhgt_reserve(g->ctx, bb + 256, (uint32_t)X0.kept_per_block, (W + 2) * X0.n_sc, X0.chunk_nbytes, g->o.typesize,
                              g->o.blocksize) == HHGT_OK;
        }
        if (ok && dev) {
            const uint64_t want = (uint64_t)((double)bb_dev / 24.0 * 1.15) + (256u << 10);
            const size_t tab_room = (size_t)(want / 26 + 2) * 28 + 64;
            for (auto &sg : g->stg) ok = ok && sg.h.ensure((size_t)want + 64 + tab_room) == HHGT_OK && sg.d.ensure((size_t)want + 64 + tab_room) == HHGT_OK;
            // per text buffer: a status word per member (files written by bgzip hold ~64 KB of text per member; four times as
            // many fit before these grow inside a pass — a hipFree there waits for the device to drain)
            for (auto &tb : g->text) ok = ok && tb.status.ensure((size_t)(bb_dev / 16384 + 64) * 4) == HHGT_OK && tb.bad.ensure(8) == HHGT_OK;
        }
        if (ok && g->o.device_inflate != 1) hhgt_reader_prewarm(bb_host, 6 * ((g->o.files_ahead > 0 ? g->o.files_ahead : 1) + 1));
        if (ok && S > 0) {
            // one chunk column through the compressor on the engine's stream: the first launch of a kernel that spills
            // (k_lz4_blocks) makes the runtime allocate the queue's scratch arena — 28 ms inside the first batch of the first
            // input otherwise (HHGT_INGEST_DEBUG: "harvest work" 28.1 ms against 0.3)
            hhgt_ingest::InState &X0 = g->ist[0];
            DstSlot &d0 = g->dst[0];
            if (X0.planes) {
                ok = hipMemsetAsync(X0.P.p, 0, (size_t)hhgt_planes_bytes(&X0.lay), g->s_main) == hipSuccess &&
                     hhgt_compress_planes(g->ctx, &X0.lay, X0.P.p, X0.G.p, 0u, 1u, g->o.format, d0.d.p, d0.d.cap, d0.off.as<uint64_t>(), nullptr,
                                          g->s_main) == HHGT_OK;
            } else {
                ok = hipMemsetAsync(X0.G.p, 0, (size_t)X0.col_bytes, g->s_main) == hipSuccess &&
                     hhgt_compress_chunks(g->ctx, X0.G.p, X0.n_sc, X0.chunk_nbytes, g->o.typesize, g->o.blocksize, g->o.format, d0.d.p, d0.d.cap,
                                          d0.off.as<uint64_t>(), nullptr, g->s_main) == HHGT_OK;
            }
            ok = ok && hipStreamSynchronize(g->s_main) == hipSuccess;
        }
        if (ok) {
            // ... and one small copy each way on every stream the engine copies on (the first copy of a direction on a stream
            // sets the runtime's copy path up: milliseconds, once)
            hhgt_ingest::InState &X0 = g->ist[0];
            DstSlot &d0 = g->dst[0];
            uint8_t *h8 = d0.h_off.p;   // pinned, >= 16 bytes
            for (hipStream_t st : {g->s_main, g->s_copy, g->s_inf, g->s_inf2, g->s_carry, g->s_out}) {
                ok = ok && hipMemcpyAsync(X0.cursor.p, h8, 8, hipMemcpyHostToDevice, st) == hipSuccess &&
                     hipMemcpyAsync(h8 + 8, X0.cursor.p, 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipMemsetAsync(X0.cursor.p, 0, 8, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            }
            // ... and every pinned buffer the device will copy into is copied into once, in full (HHGT_INGEST_DEBUG=2 showed the
            // first device -> host copy into each fresh pinned slot taking 7-10 ms where later ones take microseconds)
            if (S > 0) {
                for (auto &v : g->var) {
                    const size_t n4 = v.start.cap < X0.t_start.cap ? v.start.cap : X0.t_start.cap, n1 = v.ref.cap < X0.t_ref.cap ? v.ref.cap : X0.t_ref.cap;
                    ok = ok && hipMemcpyAsync(v.start.p, X0.t_start.p, n4, hipMemcpyDeviceToHost, g->s_main) == hipSuccess &&
                         hipMemcpyAsync(v.ref.p, X0.t_ref.p, n1, hipMemcpyDeviceToHost, g->s_main) == hipSuccess &&
                         hipMemcpyAsync(v.alt.p, X0.t_alt.p, n1, hipMemcpyDeviceToHost, g->s_main) == hipSuccess;
                }
                for (auto &d : g->dst)
                    ok = ok && hipMemcpyAsync(d.h_off.p, d.off.p, d.h_off.cap < d.off.cap ? d.h_off.cap : d.off.cap, hipMemcpyDeviceToHost, g->s_main) == hipSuccess;
                for (auto &o : g->out)
                    ok = ok && hipMemcpyAsync(o.h.p, d0.d.p, o.h.cap < d0.d.cap ? o.h.cap : d0.d.cap, hipMemcpyDeviceToHost, g->s_out) == hipSuccess;
                ok = ok && hipStreamSynchronize(g->s_main) == hipSuccess && hipStreamSynchronize(g->s_out) == hipSuccess;
            }
        }
        if (!ok) {
            hhgt_ingest_close(g);
            return HHGT_ERR_HIP;
        }
    }
    g->tm.t_open = now_s();
    g->th_source = std::thread(source_main, g);
    g->th_driver = std::thread(driver_main, g);
    g->th_ship = std::thread(ship_main, g);
    *out = g;
    return HHGT_OK;
}

static int add_input(hhgt_ingest *g, std::unique_ptr<Input> in)
{
    std::lock_guard<std::mutex> lk(g->in_mu);
    if (g->finished) {
        hhgt_set_error("ingest: inputs cannot be added after hhgt_ingest_finish");
        return HHGT_ERR_ARG;
    }
    in->index = (int)g->inputs.size();
    const int idx = in->index;
    g->inputs.push_back(std::move(in));
    g->in_cv.notify_all();
    return idx;
}

extern "C" int hhgt_ingest_add_file(hhgt_ingest *g, const char *path, const char *region)
{
    if (!g || !path) return HHGT_ERR_ARG;
    trace("add_file");
    if (access(path, R_OK) != 0) {
        hhgt_set_error("cannot open %s", path);
        return HHGT_ERR_IO;
    }
    std::unique_ptr<Input> in(new Input());
    in->kind = 0;
    in->path = path;
    in->region = region ? region : "";
    return add_input(g, std::move(in));
}

extern "C" int hhgt_ingest_add_memory(hhgt_ingest *g, const void *host_text, uint64_t nbytes, const char *region)
{
    if (!g || (!host_text && nbytes)) return HHGT_ERR_ARG;
    std::unique_ptr<Input> in(new Input());
    in->kind = 1;
    in->mem = static_cast<const uint8_t *>(host_text);
    in->mem_bytes = nbytes;
    in->region = region ? region : "";
    return add_input(g, std::move(in));
}

extern "C" int hhgt_ingest_finish(hhgt_ingest *g)
{
    if (!g) return HHGT_ERR_ARG;
    {
        std::lock_guard<std::mutex> lk(g->in_mu);
        g->finished = true;
    }
    g->in_cv.notify_all();
    return HHGT_OK;
}

static void release_held(hhgt_ingest *g)
{
    if (!g->have_held) return;
    if (g->held.var_slot >= 0) g->free_var.push(g->held.var_slot);
    if (g->held.out_slot >= 0) g->free_out.push(g->held.out_slot);
    g->have_held = false;
}

extern "C" int hhgt_ingest_next(hhgt_ingest *g, hhgt_ingest_event *ev)
{
    if (!g || !ev) return HHGT_ERR_ARG;
    memset(ev, 0, sizeof(*ev));
    release_held(g);
    if (g->ended) return HHGT_OK;   // kind stays HHGT_EV_END
    Batch b;
    if (!g->q_out.pop(b)) {
        std::lock_guard<std::mutex> lk(g->err_mu);
        hhgt_set_error("%s", g->errmsg.empty() ? "ingest: stopped" : g->errmsg.c_str());
        g->ended = true;
        return g->err ? g->err : HHGT_ERR_IO;
    }
    g->held = b;
    g->have_held = true;
    ev->kind = b.kind;
    ev->input = b.in ? b.in->index : -1;
    switch (b.kind) {
    case B_HEADER:
        ev->header = b.in->header.data();
        ev->header_bytes = b.in->header.size();
        ev->n_samples = b.in->S;
        break;
    case B_VARIANTS: {
        VarSlot &v = g->var[(size_t)b.var_slot];
        ev->start = reinterpret_cast<const uint32_t *>(v.start.p);
        ev->ref = v.ref.p;
        ev->alt = v.alt.p;
        ev->first_variant = b.first_variant;
        ev->n_variants = b.n_variants;
        ev->n_runs = b.n_runs;
        ev->run_first = v.run_first;
        ev->run_names = &v.run_names[0][0];
        break;
    }
    case B_COLUMNS: {
        OutSlot &o = g->out[(size_t)b.out_slot];
        ev->framed = o.h.p;
        ev->chunk_off = o.off.data();
        ev->framed_bytes = b.framed_bytes;
        ev->n_chunks = b.n_chunks;
        ev->first_col = b.first_col;
        ev->n_cols = b.n_cols;
        ev->raw_bytes = b.raw_bytes;
        break;
    }
    case B_INPUT_END:
        ev->stats = b.in->st;
        break;
    default:
        g->ended = true;
        break;
    }
    return HHGT_OK;
}

// The event returned by the last hhgt_ingest_next keeps its buffers past the next call: the consumer writes a batch of
// chunks to a file on another thread while it already takes the next event (round 4: the converter's run was the engine's
// 0.2 s stretched to 0.7 by the consumer's writes).  The token names the slots; hhgt_ingest_release may come from any thread.
extern "C" int hhgt_ingest_hold(hhgt_ingest *g, int *token)
{
    if (!g || !token) return HHGT_ERR_ARG;
    if (!g->have_held) {
        hhgt_set_error("hhgt_ingest_hold: no event is held (call it after hhgt_ingest_next, once per event)");
        return HHGT_ERR_ARG;
    }
    *token = (g->held.out_slot + 1) | ((g->held.var_slot + 1) << 8);
    g->have_held = false;
    return HHGT_OK;
}

extern "C" int hhgt_ingest_release(hhgt_ingest *g, int token)
{
    if (!g || token < 0) return HHGT_ERR_ARG;
    const int out = (token & 0xFF) - 1, var = ((token >> 8) & 0xFF) - 1;
    if (out >= N_OUT || var >= N_VAR) return HHGT_ERR_ARG;
    if (var >= 0) g->free_var.push(var);
    if (out >= 0) g->free_out.push(out);
    return HHGT_OK;
}

extern "C" void hhgt_ingest_close(hhgt_ingest *g)
{
    if (!g) return;
    hipSetDevice(g->device);
    // stop the stages (a normal end has already drained them)
    if (!g->ended) fail(g, HHGT_ERR_IO, "ingest: closed");
    hhgt_ingest_finish(g);
    if (g->th_source.joinable()) g->th_source.join();
    if (g->th_driver.joinable()) g->th_driver.join();
    if (g->th_ship.joinable()) g->th_ship.join();
    if (trace_level() >= 2) {
        std::lock_guard<std::mutex> lk(g_trace_mu);
        const double t0 = g_trace.empty() ? 0 : g_trace[0].t;
        fprintf(stderr, "[trace] t0 = %.3f ms (steady clock; HHGT_ALLOC_DEBUG lines carry the same clock)\n", t0 * 1e3);
        for (auto &r : g_trace) fprintf(stderr, "[trace] %9.3f ms  %-20s %lld %lld\n", (r.t - t0) * 1e3, r.tag, r.a, r.b);
        g_trace.clear();
    }
    for (hipStream_t s : {g->s_main, g->s_copy, g->s_inf, g->s_inf2, g->s_carry, g->s_out})
        if (s) {
            hipStreamSynchronize(s);
            hipStreamDestroy(s);
        }
    for (auto &tb : g->text) {
        if (tb.d) hipFree(tb.d);
        if (tb.ready) hipEventDestroy(tb.ready);
        if (tb.carry_done) hipEventDestroy(tb.carry_done);
        if (tb.h_bad) hipHostFree(tb.h_bad);
        tb.status.release();
        tb.bad.release();
    }
    for (auto &s : g->stg) {
        s.h.release();
        s.d.release();
        if (s.done) hipEventDestroy(s.done);
    }
    g->crc_x2n.release();
    for (auto &x : g->ist)
        for (DevBuf *b : {&x.G, &x.P, &x.t_start, &x.t_ref, &x.t_alt, &x.cursor}) b->release();
    for (auto &r : g->res) {
        if (r.rec) hipHostFree(r.rec);
        if (r.ev) hipEventDestroy(r.ev);
    }
    for (auto &v : g->var) {
        v.start.release();
        v.ref.release();
        v.alt.release();
    }
    for (auto &d : g->dst) {
        d.d.release();
        d.off.release();
        d.h_off.release();
    }
    for (auto &o : g->out) o.h.release();
    for (hipEvent_t e : g->batch_events) hipEventDestroy(e);
    {
        std::lock_guard<std::mutex> lk(g->in_mu);
        for (auto &in : g->inputs)
            if (in->rd) hhgt_reader_close(in->rd);
    }
    delete g;
}
