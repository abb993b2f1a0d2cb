// api.hip — C ABI of libhhgt.so (see include/hhgt.h): context, orchestration of the kernel stages,
// error reporting.  No CPU fallback anywhere: every compute entry point needs a HIP device.
#include "common.h"
#include <chrono>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void hhgt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *hhgt_last_error(void) { return g_err; }
extern "C" const char *hhgt_version(void) { return "hhgt 0.1 (gfx950)"; }

extern "C" int hhgt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return HHGT_OK;
    if (p) {
        hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    static const bool dbg = getenv("HHGT_ALLOC_DEBUG") != nullptr;   // development aid: late allocations stall every stream
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, want);
    if (dbg)
        fprintf(stderr, "[alloc] %.3f ms  hipMalloc %zu took %.3f ms\n", std::chrono::duration<double>(t0.time_since_epoch()).count() * 1e3,
                want, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    if (e != hipSuccess) {
        p = nullptr;
        hhgt_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return HHGT_ERR_HIP;
    }
    cap = want;
    return HHGT_OK;
}

void DevBuf::release()
{
    if (p) hipFree(p);
    p = nullptr;
    cap = 0;
}

#define TRY(expr)                      \
    do {                               \
        int _rc = (expr);              \
        if (_rc != HHGT_OK) return _rc; \
    } while (0)

extern "C" int hhgt_ctx_create(int device, hhgt_ctx **out)
{
    if (!out) return HHGT_ERR_ARG;
    *out = nullptr;
    int n = hhgt_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        hhgt_set_error("no usable HIP device (count=%d, requested=%d): libhhgt has no CPU path", n, device);
        return HHGT_ERR_NO_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    hhgt_ctx *c = new hhgt_ctx();
    c->device = device;
    // from here on a failure must not leak the context (hhgt_ctx_destroy copes with a half-built one)
    int rc = HHGT_OK;
    hipError_t e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) {
        hhgt_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        rc = HHGT_ERR_HIP;
    } else if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        hhgt_set_error("device %d is %s; libhhgt carries gfx950 code objects only", device, c->prop.gcnArchName);
        rc = HHGT_ERR_NO_DEVICE;
    }
    if (rc == HHGT_OK && (e = hipHostMalloc(reinterpret_cast<void **>(&c->h_counters), sizeof(DevCounters), hipHostMallocDefault)) != hipSuccess) {
        hhgt_set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
        rc = HHGT_ERR_HIP;
    }
    if (rc == HHGT_OK && (e = hipHostMalloc(reinterpret_cast<void **>(&c->h_result_pinned), sizeof(hhgt_encode_result), hipHostMallocDefault)) != hipSuccess) {
        hhgt_set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
        rc = HHGT_ERR_HIP;
    }
    if (rc == HHGT_OK) rc = c->counters.ensure(sizeof(DevCounters));
    if (rc != HHGT_OK) {
        hhgt_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return HHGT_OK;
}

extern "C" void hhgt_ctx_destroy(hhgt_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    DevBuf *bufs[] = {&c->slots, &c->counts, &c->prefix, &c->nl, &c->scan_tmp, &c->l_soff, &c->l_lend, &c->l_pos,
                      &c->l_refalt, &c->l_flags, &c->l_keep, &c->l_kidx, &c->l_cnew, &c->l_crun, &c->k_soff,
                      &c->k_lend, &c->k_meta, &c->redo_list, &c->redo_flag, &c->run_first, &c->run_names,
                      &c->counters, &c->cursor, &c->result, &c->dec_bad, &c->oh_ovl, &c->oh_lut, &c->crc_x2n};
    for (DevBuf *b : bufs) b->release();
    for (auto &w : c->cw) {
        DevBuf *wb[] = {&w.lz_scratch, &w.lz_csize, &w.lz_flags, &w.fr_bsize, &w.fr_csize, &w.fr_flags, &w.fr_state};
        for (DevBuf *b : wb) b->release();
        if (w.lz_done) hipEventDestroy(w.lz_done);
        if (w.fr_done) hipEventDestroy(w.fr_done);
    }
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->h_result_pinned) hipHostFree(c->h_result_pinned);
    for (auto &p : c->pending) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    for (auto e : c->event_pool) hipEventDestroy(e);
    delete c;
}

// ---- profiling ------------------------------------------------------------------------------------
static hipEvent_t get_event(hhgt_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

StageTimer::StageTimer(hhgt_ctx *c, hipStream_t s, int stage_) : ctx(c), st(s), stage(stage_)
{
    if (ctx->profiling) {
        a = get_event(ctx);
        b = get_event(ctx);
        hipEventRecord(a, st);
    }
}

void StageTimer::stop()
{
    if (a) {
        hipEventRecord(b, st);
        ctx->pending.push_back({stage, a, b});
        a = b = nullptr;
    }
}

int hhgt_profile_collect(hhgt_ctx *c)
{
    for (auto &p : c->pending) {
        hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->stage_ms[p.stage] += ms;
            c->stage_launches[p.stage] += 1;
        }
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->pending.clear();
    return HHGT_OK;
}

extern "C" int hhgt_profile_enable(hhgt_ctx *c, int on)
{
    if (!c) return HHGT_ERR_ARG;
    c->profiling = on ? 1 : 0;
    return HHGT_OK;
}

extern "C" int hhgt_profile_reset(hhgt_ctx *c)
{
    if (!c) return HHGT_ERR_ARG;
    hhgt_profile_collect(c);
    for (int i = 0; i < HHGT_N_STAGES; ++i) {
        c->stage_ms[i] = 0;
        c->stage_launches[i] = 0;
    }
    return HHGT_OK;
}

extern "C" int hhgt_profile_read(hhgt_ctx *c, double *ms, uint64_t *launches)
{
    if (!c) return HHGT_ERR_ARG;
    hhgt_profile_collect(c);
    for (int i = 0; i < HHGT_N_STAGES; ++i) {
        if (ms) ms[i] = c->stage_ms[i];
        if (launches) launches[i] = c->stage_launches[i];
    }
    return HHGT_OK;
}

// ---- layout ---------------------------------------------------------------------------------------
static int make_layout(const hhgt_layout *lay, LayoutDev *L)
{
    if (!lay || lay->n_samples < 0) {
        hhgt_set_error("layout: bad n_samples");
        return HHGT_ERR_ARG;
    }
    L->S = (uint32_t)lay->n_samples;
    L->v_capacity = lay->v_capacity;
    L->ring = 0;
    L->pad_ = 0;
    if (lay->ring < 0 || (lay->ring > 0 && (lay->vc <= 0 || lay->v_capacity != (uint64_t)lay->ring * (uint64_t)lay->vc))) {
        hhgt_set_error("layout: a ring needs vc > 0 and v_capacity == ring * vc");
        return HHGT_ERR_ARG;
    }
    L->ring = (uint32_t)lay->ring;
    if (lay->vc == 0) {
        L->Vc = lay->v_capacity ? lay->v_capacity : TILE_V;
    } else {
        L->Vc = (uint64_t)lay->vc;
    }
    if (L->Vc % TILE_V) {
        hhgt_set_error("layout: variants per chunk (%llu) must be a multiple of %d (dense: v_capacity)",
                       (unsigned long long)L->Vc, TILE_V);
        return HHGT_ERR_ARG;
    }
    if (lay->v_capacity % L->Vc) {
        hhgt_set_error("layout: v_capacity must be a multiple of vc");
        return HHGT_ERR_ARG;
    }
    if (lay->sc == 0) {
        L->Sc = L->S ? L->S : 1;
        L->sc_log2 = 31;
        L->n_sc = 1;
    } else {
        uint32_t sc = (uint32_t)lay->sc;
        if (sc & (sc - 1)) {
            hhgt_set_error("layout: sc must be a power of two");
            return HHGT_ERR_ARG;
        }
        L->Sc = sc;
        uint32_t lg = 0;
        while ((1u << lg) < sc) ++lg;
        L->sc_log2 = lg;
        L->n_sc = (L->S + sc - 1) / sc;
        if (L->n_sc == 0) L->n_sc = 1;
    }
    return HHGT_OK;
}

extern "C" uint64_t hhgt_layout_bytes(const hhgt_layout *lay)
{
    LayoutDev L;
    if (make_layout(lay, &L) != HHGT_OK) return 0;
    return (uint64_t)L.n_sc * L.Sc * L.v_capacity * 2ull;
}

extern "C" uint64_t hhgt_layout_offset(const hhgt_layout *lay, uint32_t s, uint64_t v)
{
    LayoutDev L;
    if (make_layout(lay, &L) != HHGT_OK) return ~0ull;
    return layout_offset(L, s, v);
}

static int parse_region_host(const char *region, RegionFilter *rf)
{
    memset(rf, 0, sizeof(*rf));
    if (!region || !*region) return HHGT_OK;
    const char *colon = strrchr(region, ':');
    size_t clen = strlen(region);
    if (colon && colon[1] >= '0' && colon[1] <= '9') {
        char *p;
        long long b = strtoll(colon + 1, &p, 10), e = 0x7fffffffffffffffLL;
        while (*p == ',') p++;
        if (*p == '-' && p[1]) e = strtoll(p + 1, &p, 10);
        clen = (size_t)(colon - region);
        rf->has_range = 1;
        rf->beg = b;
        rf->end = e;
    }
    if (clen == 0 || clen >= sizeof(rf->contig)) {
        hhgt_set_error("region '%s': contig name empty or longer than 63 bytes", region);
        return HHGT_ERR_ARG;
    }
    memcpy(rf->contig, region, clen);
    rf->contig_len = (int)clen;
    return HHGT_OK;
}

// ---- encode ---------------------------------------------------------------------------------------
// device-side epilogue of one encode call: advance the cursor, assemble the result record (counts, CHROM runs)
__global__ void k_encode_finish(DevCounters *cnt, uint64_t *cursor, const uint64_t *__restrict__ run_first,
                                const uint8_t *__restrict__ run_names, uint64_t v_capacity, hhgt_encode_result *res)
{
    const uint32_t t = threadIdx.x;
    const uint64_t before = *cursor;
    const uint64_t kept = cnt->n_kept;
    __syncthreads();
    if (t == 0) {
        *cursor = before + kept;
        cnt->cursor_after = before + kept;
        res->stats.n_lines = cnt->n_lines;
        res->stats.n_records = cnt->n_records;
        res->stats.n_kept = kept;
        res->stats.n_drop_region = cnt->n_drop_region;
        res->stats.n_drop_filter = cnt->n_drop_filter;
        res->stats.n_haploid_padded = cnt->n_haploid;
        res->stats.n_malformed = cnt->n_malformed;
        res->stats.n_general_lines = cnt->n_general;
        res->stats.n_chrom_runs = cnt->n_chrom_runs;
        res->cursor_before = before;
        res->cursor_after = before + kept;
        res->n_lines_over = cnt->err_lines;
        res->err_density = cnt->err_density;
        res->v_capacity = v_capacity;
        res->done = 1u;
        res->reserved = cnt->n_other > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt->n_other;   // bit-plane form: calls beyond 0 / 1 / missing
    }
    const uint64_t n_runs = cnt->n_chrom_runs;
    if (t < HHGT_RESULT_RUNS) res->run_first[t] = t < n_runs ? run_first[t] : 0ull;
    for (uint32_t q = t; q < HHGT_RESULT_RUNS * 32u; q += blockDim.x)
        res->run_names[q >> 5][q & 31u] = (q >> 5) < n_runs ? (char)run_names[q] : 0;
    // the counters go back to zero for the next call on this context (one memset launch less per call: counters_clean)
    __syncthreads();
    if (t < sizeof(DevCounters) / 8u) reinterpret_cast<unsigned long long *>(cnt)[t] = 0ull;
}

__global__ void k_set_u64(uint64_t *p, uint64_t v) { *p = v; }

static int encode_check_args(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const hhgt_layout *lay, LayoutDev *L,
                             const char *region, RegionFilter *rf, void *d_G)
{
    TRY(make_layout(lay, L));
    if (nbytes >= 0xFFFFFFF0ull) {
        hhgt_set_error("encode: text block of %llu bytes; split it below 4 GiB at a line boundary",
                       (unsigned long long)nbytes);
        return HHGT_ERR_ARG;
    }
    if (nbytes && (!d_text || (reinterpret_cast<uintptr_t>(d_text) & 15u))) {
        hhgt_set_error("encode: d_text must be a 16-byte aligned device pointer");
        return HHGT_ERR_ARG;
    }
    if (L->S > 0 && !d_G) {
        hhgt_set_error("encode: d_G is NULL");
        return HHGT_ERR_ARG;
    }
    const int rc = parse_region_host(region, rf);
    rf->keep_multi = c->keep_multi;
    return rc;
}

// stage 1: newline index of the block -> prefix[n_regions] holds the line count (device).  min_line: a record with S
// sample columns cannot be shorter than 9 fixed columns and S fields of one byte and a separator each (0: unknown)
static int encode_stage_index(hhgt_ctx *c, const uint8_t *text, uint64_t nbytes, uint32_t n_regions, uint32_t min_line, uint32_t S,
                              DevCounters *cnt, hipStream_t st)
{
    TRY(c->slots.ensure((size_t)n_regions * INDEX_CAP * 4));
    TRY(c->counts.ensure((size_t)n_regions * 4));
    TRY(c->prefix.ensure(((size_t)n_regions + 1) * 4));
    TRY(c->scan_tmp.ensure(2 * scan_tmp_elems(n_regions) * 4));
    TRY(launch_index_newlines(text, nbytes, c->slots.as<uint32_t>(), c->counts.as<uint32_t>(), n_regions, min_line, S, c->index_mode, cnt, st));
    TRY(launch_scan_exclusive_u32_pair(c->counts.as<uint32_t>(), c->prefix.as<uint32_t>(), nullptr, nullptr, n_regions,
                                       c->scan_tmp.as<uint32_t>(), c->scan_tmp.cap / 4, st));
    return HHGT_OK;
}

// HHGT_ENC_STRIDE=0: GT:DP-style records of one column width go to the variable-width kernel like every other record that is not
// "a|b\t" x S (round 3); default: the bit-plane tile kernel decodes them at their stride
static bool stride_env()
{
    static const bool on = !(getenv("HHGT_ENC_STRIDE") && atoi(getenv("HHGT_ENC_STRIDE")) == 0);
    return on;
}

// stages 2..: everything behind the index, sized by max_lines, counts and append position read on the device
static int encode_stage_rest(hhgt_ctx *c, const uint8_t *text, uint64_t nbytes, uint32_t n_regions, uint32_t max_lines,
                             const RegionFilter &rf, const LayoutDev &L, const uint64_t *d_cursor, void *d_G, void *d_P, uint32_t *d_start, uint32_t *d_stop,
                             uint8_t *d_ref, uint8_t *d_alt, DevCounters *cnt, hipStream_t st)
{
    const uint32_t *d_nlines = c->prefix.as<uint32_t>() + n_regions;
    const size_t nl4 = ((size_t)max_lines + 1) * 4;
    TRY(c->nl.ensure(nl4));
    TRY(c->scan_tmp.ensure(2 * scan_tmp_elems(max_lines) * 4));
    DevBuf *per_line[] = {&c->l_soff, &c->l_lend, &c->l_pos, &c->l_refalt, &c->l_flags, &c->l_keep,
                          &c->l_kidx, &c->l_cnew, &c->l_crun, &c->k_soff, &c->k_lend, &c->k_meta,
                          &c->redo_list, &c->redo_flag};
    for (DevBuf *b : per_line) TRY(b->ensure(nl4));
    TRY(c->run_first.ensure(MAX_CHROM_RUNS * 8));
    TRY(c->run_names.ensure(MAX_CHROM_RUNS * 32));
    {
        StageTimer t(c, st, HHGT_STAGE_INDEX);
        TRY(launch_compact_newlines(c->slots.as<uint32_t>(), c->counts.as<uint32_t>(), c->prefix.as<uint32_t>(),
                                    n_regions, c->nl.as<uint32_t>(), max_lines, st));
        t.stop();
    }
    if (max_lines == 0) return HHGT_OK;
    {
        StageTimer t(c, st, HHGT_STAGE_FIXED);
        TRY(launch_parse_fixed(text, nbytes, c->nl.as<uint32_t>(), d_nlines, max_lines, rf, L.S,
                               c->l_soff.as<uint32_t>(), c->l_lend.as<uint32_t>(), c->l_pos.as<uint32_t>(),
                               c->l_refalt.as<uint32_t>(), c->l_flags.as<uint32_t>(), c->l_keep.as<uint32_t>(),
                               c->l_cnew.as<uint32_t>(), c->index_mode, cnt, st));
        TRY(launch_scan_exclusive_u32_pair(c->l_keep.as<uint32_t>(), c->l_kidx.as<uint32_t>(), c->l_cnew.as<uint32_t>(),
                                           c->l_crun.as<uint32_t>(), max_lines, c->scan_tmp.as<uint32_t>(), c->scan_tmp.cap / 4, st));
        TRY(launch_compact_kept(text, nbytes, c->nl.as<uint32_t>(), d_nlines, max_lines, c->l_soff.as<uint32_t>(),
                                c->l_lend.as<uint32_t>(), c->l_pos.as<uint32_t>(), c->l_refalt.as<uint32_t>(),
                                c->l_flags.as<uint32_t>(), c->l_kidx.as<uint32_t>(), c->l_crun.as<uint32_t>(),
                                c->k_soff.as<uint32_t>(), c->k_lend.as<uint32_t>(), c->k_meta.as<uint32_t>(),
                                c->redo_list.as<uint32_t>(), c->redo_flag.as<uint32_t>(), c->run_first.as<uint64_t>(), c->run_names.as<uint8_t>(),
                                MAX_CHROM_RUNS, d_cursor, L.v_capacity, L.ring, d_start, d_stop, d_ref, d_alt, cnt, d_P != nullptr && stride_env(), st));
        t.stop();
    }
    if (L.S > 0) {
        StageTimer t(c, st, HHGT_STAGE_ENCODE);
        if (d_P)
            TRY(launch_encode_planes(text, nbytes, c->k_soff.as<uint32_t>(), c->k_meta.as<uint32_t>(), max_lines, d_cursor, L,
                                     static_cast<uint8_t *>(d_P), static_cast<int8_t *>(d_G), c->redo_list.as<uint32_t>(),
                                     c->redo_flag.as<uint32_t>(), cnt, st));
        else
            TRY(launch_encode_tiles(text, nbytes, c->k_soff.as<uint32_t>(), c->k_meta.as<uint32_t>(), max_lines, d_cursor,
                                    L, static_cast<int8_t *>(d_G), c->redo_list.as<uint32_t>(),
                                    c->redo_flag.as<uint32_t>(), cnt, st));
        t.stop();
        StageTimer t2(c, st, HHGT_STAGE_GENERAL);
        TRY(launch_encode_general(text, nbytes, c->k_soff.as<uint32_t>(), c->k_lend.as<uint32_t>(),
                                  c->k_meta.as<uint32_t>(), c->redo_list.as<uint32_t>(), d_cursor, L,
                                  static_cast<int8_t *>(d_G), static_cast<uint8_t *>(d_P), cnt, c->prop.multiProcessorCount, st));
        t2.stop();
    }
    return HHGT_OK;
}

static int encode_finish(hhgt_ctx *c, uint64_t *d_cursor, const LayoutDev &L, DevCounters *cnt, hipStream_t st)
{
    TRY(c->run_first.ensure(MAX_CHROM_RUNS * 8));
    TRY(c->run_names.ensure(MAX_CHROM_RUNS * 32));
    TRY(c->result.ensure(sizeof(hhgt_encode_result)));
    hipLaunchKernelGGL(k_encode_finish, dim3(1), dim3(256), 0, st, cnt, d_cursor, c->run_first.as<uint64_t>(),
                       c->run_names.as<uint8_t>(), L.ring ? 0ull : L.v_capacity, c->result.as<hhgt_encode_result>());
    HIP_TRY(hipGetLastError());
    return HHGT_OK;
}

extern "C" int hhgt_encode_result_status(const hhgt_encode_result *r)
{
    if (!r) return HHGT_ERR_ARG;
    if (!r->done) {
        hhgt_set_error("encode: the result record is not complete yet (synchronise the stream first)");
        return HHGT_ERR_ARG;
    }
    if (r->err_density) {
        hhgt_set_error("encode: more than %u newlines inside one %u-byte region (not VCF text)", INDEX_CAP, INDEX_REGION);
        return HHGT_ERR_LINE_DENSITY;
    }
    if (r->n_lines_over) {
        hhgt_set_error("encode: %llu lines beyond max_lines were not encoded", (unsigned long long)r->n_lines_over);
        return HHGT_ERR_CAPACITY;
    }
    if (r->stats.n_malformed) {
        hhgt_set_error("Error parsing VCF file: %llu malformed record(s) (too few columns, bad POS, FORMAT without GT)",
                       (unsigned long long)r->stats.n_malformed);
        return HHGT_ERR_MALFORMED;
    }
    if (r->v_capacity && r->cursor_after > r->v_capacity) {
        hhgt_set_error("encode: %llu kept records at v_base %llu exceed v_capacity %llu",
                       (unsigned long long)r->stats.n_kept, (unsigned long long)r->cursor_before,
                       (unsigned long long)r->v_capacity);
        return HHGT_ERR_CAPACITY;
    }
    if (r->stats.n_chrom_runs > MAX_CHROM_RUNS) {
        hhgt_set_error("encode: %llu CHROM runs in one text block (limit %u): the input is not sorted by contig",
                       (unsigned long long)r->stats.n_chrom_runs, MAX_CHROM_RUNS);
        return HHGT_ERR_CAPACITY;
    }
    return HHGT_OK;
}

static void empty_result(hhgt_encode_result *r)
{
    memset(r, 0, sizeof(*r));
    r->done = 1u;
}

// a layout can carry bit planes when a Blosc block (4096 variants of one sample) never crosses a chunk column
static int planes_check_layout(const LayoutDev &L)
{
    if (L.Vc % 4096ull) {
        hhgt_set_error("planes: variants per chunk (%llu; dense: v_capacity) must be a multiple of 4096", (unsigned long long)L.Vc);
        return HHGT_ERR_ARG;
    }
    return HHGT_OK;
}

extern "C" uint64_t hhgt_planes_bytes(const hhgt_layout *lay)
{
    LayoutDev L;
    if (make_layout(lay, &L) != HHGT_OK || planes_check_layout(L) != HHGT_OK) return 0;
    return (uint64_t)L.n_sc * L.Sc * L.v_capacity / 2ull;
}

// columns [col0, col0 + n_cols) of a plane buffer: checks and geometry shared by compress and expand
static int planes_columns(const hhgt_layout *lay, uint32_t col0, uint32_t n_cols, LayoutDev *L, PlanesGeom *pg)
{
    TRY(make_layout(lay, L));
    TRY(planes_check_layout(*L));
    if ((uint64_t)col0 + n_cols > L->v_capacity / L->Vc) {
        hhgt_set_error("planes: columns [%u, %u) outside the layout's %llu", col0, col0 + n_cols, (unsigned long long)(L->v_capacity / L->Vc));
        return HHGT_ERR_ARG;
    }
    *pg = planes_geom(*L, col0);
    return HHGT_OK;
}

static int encode_async_impl(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region, const hhgt_layout *lay,
                             uint64_t *d_cursor, uint32_t max_lines, void *d_G, void *d_P, bool planes, uint32_t *d_start,
                             uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, hhgt_encode_result *h_result, void *stream)
{
    if (!c || !d_cursor) return HHGT_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    RegionFilter rf;
    TRY(encode_check_args(c, d_text, nbytes, lay, &L, region, &rf, planes ? d_P : d_G));
    if (planes) TRY(planes_check_layout(L));
    const uint8_t *text = static_cast<const uint8_t *>(d_text);
    DevCounters *cnt = c->counters.as<DevCounters>();
    if (!c->counters_clean) HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(DevCounters), st));   // (the last call's epilogue did not get to zero them)
    c->counters_clean = false;
    const uint32_t n_regions = (uint32_t)((nbytes + 1 + INDEX_REGION - 1) / INDEX_REGION);
    if (nbytes) {
        {
            StageTimer t(c, st, HHGT_STAGE_INDEX);
            TRY(encode_stage_index(c, text, nbytes, n_regions, L.S ? 2u * L.S + 17u : 0u, L.S, cnt, st));
            t.stop();
        }
        TRY(encode_stage_rest(c, text, nbytes, n_regions, max_lines, rf, L, d_cursor, d_G, planes ? d_P : nullptr, d_start, d_stop, d_ref,
                              d_alt, cnt, st));
    }
    TRY(encode_finish(c, d_cursor, L, cnt, st));
    c->counters_clean = true;   // (calls on one context are serialised by the caller: the next one is queued behind this epilogue)
    if (h_result) {
        h_result->done = 0u;
        HIP_TRY(hipMemcpyAsync(h_result, c->result.p, sizeof(hhgt_encode_result), hipMemcpyDeviceToHost, st));
    }
    return HHGT_OK;
}

extern "C" int hhgt_encode_text_async(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region,
                                      const hhgt_layout *lay, uint64_t *d_cursor, uint32_t max_lines, void *d_G,
                                      uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt,
                                      hhgt_encode_result *h_result, void *stream)
{
    return encode_async_impl(c, d_text, nbytes, region, lay, d_cursor, max_lines, d_G, nullptr, false, d_start, d_stop, d_ref, d_alt,
                             h_result, stream);
}

extern "C" int hhgt_encode_text_planes_async(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region,
                                             const hhgt_layout *lay, uint64_t *d_cursor, uint32_t max_lines, void *d_P, void *d_G,
                                             uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt,
                                             hhgt_encode_result *h_result, void *stream)
{
    return encode_async_impl(c, d_text, nbytes, region, lay, d_cursor, max_lines, d_G, d_P, true, d_start, d_stop, d_ref, d_alt,
                             h_result, stream);
}

extern "C" int hhgt_pad_tail_planes(hhgt_ctx *c, const hhgt_layout *lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end,
                                    void *d_P, void *stream)
{
    if (!c || !d_P) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    TRY(make_layout(lay, &L));
    TRY(planes_check_layout(L));
    if (v_end > L.v_capacity || vcol_end > L.v_capacity / L.Vc) {
        hhgt_set_error("pad_tail: range outside the layout");
        return HHGT_ERR_ARG;
    }
    return launch_pad_tail_planes(L, v_end, vcol_begin, vcol_end, static_cast<uint8_t *>(d_P), reinterpret_cast<hipStream_t>(stream));
}

extern "C" int hhgt_pad_tail_planes_cursor(hhgt_ctx *c, const hhgt_layout *lay, const uint64_t *d_cursor, void *d_P, void *stream)
{
    if (!c || !d_P || !d_cursor) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    TRY(make_layout(lay, &L));
    TRY(planes_check_layout(L));
    return launch_pad_tail_planes_cursor(L, d_cursor, static_cast<uint8_t *>(d_P), reinterpret_cast<hipStream_t>(stream));
}

extern "C" int hhgt_planes_expand(hhgt_ctx *c, const hhgt_layout *lay, const void *d_P, const void *d_G, uint32_t col0, uint32_t n_cols,
                                  void *d_out, void *stream)
{
    if (!c || !d_P || !d_out) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    PlanesGeom pg;
    TRY(planes_columns(lay, col0, n_cols, &L, &pg));
    if (n_cols == 0) return HHGT_OK;
    return launch_planes_expand(L, static_cast<const uint8_t *>(d_P), static_cast<const uint8_t *>(d_G), col0, n_cols,
                                static_cast<uint8_t *>(d_out), reinterpret_cast<hipStream_t>(stream));
}

static int encode_text_once(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region, const hhgt_layout *lay, uint64_t v_base,
                            void *d_G, uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, hhgt_encode_stats *stats,
                            void *stream);

// The synchronous form decodes whatever the plain scan decodes: when the call comes back MALFORMED and the line index of long
// records skipped bytes (hop / walk, S >= 760), the reason may be a newline in the part it did not look at — a record shorter
// than its S samples that is valid VCF all the same (rows of empty columns; one haploid call + one two-digit allele in front of
// an empty line: DESIGN.md 4).  Then the call runs once more with every byte scanned (mode 0) and THAT outcome stands: an
// error only where the scan errs too.  Costs nothing unless the first pass fails.  (The asynchronous form reports the first
// pass's error — its callers have queued more work behind it.)
extern "C" int hhgt_encode_text(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region,
                                const hhgt_layout *lay, uint64_t v_base, void *d_G, uint32_t *d_start,
                                uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, hhgt_encode_stats *stats,
                                void *stream)
{
    int rc = encode_text_once(c, d_text, nbytes, region, lay, v_base, d_G, d_start, d_stop, d_ref, d_alt, stats, stream);
    if (rc == HHGT_ERR_MALFORMED && c && lay && lay->n_samples >= 760) {
        const int mode = c->index_mode < 0 ? index_mode_default() : c->index_mode;
        if (mode > 0) {
            const int saved = c->index_mode;
            c->index_mode = 0;
            rc = encode_text_once(c, d_text, nbytes, region, lay, v_base, d_G, d_start, d_stop, d_ref, d_alt, stats, stream);
            c->index_mode = saved;
        }
    }
    return rc;
}

static int encode_text_once(hhgt_ctx *c, const void *d_text, uint64_t nbytes, const char *region, const hhgt_layout *lay, uint64_t v_base,
                            void *d_G, uint32_t *d_start, uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, hhgt_encode_stats *stats,
                            void *stream)
{
    if (!c) return HHGT_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    LayoutDev L;
    RegionFilter rf;
    TRY(encode_check_args(c, d_text, nbytes, lay, &L, region, &rf, d_G));
    c->run_first_kept.clear();
    c->run_names_host.clear();
    if (nbytes == 0) return HHGT_OK;

    const uint8_t *text = static_cast<const uint8_t *>(d_text);
    DevCounters *cnt = c->counters.as<DevCounters>();
    TRY(c->cursor.ensure(8));
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(DevCounters), st));
    c->counters_clean = false;
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, c->cursor.as<uint64_t>(), v_base);
    const uint32_t n_regions = (uint32_t)((nbytes + 1 + INDEX_REGION - 1) / INDEX_REGION);
    uint32_t n_lines = 0;
    {
        // the synchronous form sizes everything by the exact line count: it is read back once (any input, however
        // short its lines, fits), where the asynchronous form takes the caller's bound
        StageTimer t(c, st, HHGT_STAGE_INDEX);
        TRY(encode_stage_index(c, text, nbytes, n_regions, L.S ? 2u * L.S + 17u : 0u, L.S, cnt, st));
        c->h_counters->n_lines = 0;
        c->h_counters->err_density = 0;
        HIP_TRY(hipMemcpyAsync(&c->h_counters->n_lines, c->prefix.as<uint32_t>() + n_regions, 4,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&c->h_counters->err_density, &cnt->err_density, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        t.stop();
        n_lines = (uint32_t)(c->h_counters->n_lines & 0xFFFFFFFFull);
        if (c->h_counters->err_density) {
            hhgt_set_error("encode: more than %u newlines inside one %u-byte region (not VCF text)", INDEX_CAP,
                           INDEX_REGION);
            return HHGT_ERR_LINE_DENSITY;
        }
    }
    if (stats) stats->n_lines = n_lines;
    if (n_lines == 0) return HHGT_OK;
    TRY(encode_stage_rest(c, text, nbytes, n_regions, n_lines, rf, L, c->cursor.as<uint64_t>(), d_G, nullptr, d_start, d_stop, d_ref,
                          d_alt, cnt, st));
    TRY(encode_finish(c, c->cursor.as<uint64_t>(), L, cnt, st));
    c->counters_clean = true;
    hhgt_encode_result *hr = &c->h_result;
    hr->done = 0u;
    HIP_TRY(hipMemcpyAsync(c->h_result_pinned, c->result.p, sizeof(hhgt_encode_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *hr = *c->h_result_pinned;
    if (stats) *stats = hr->stats;
    TRY(hhgt_encode_result_status(hr));
    // CHROM runs: names were copied out of the text by the compaction kernel
    const uint32_t n_runs = (uint32_t)hr->stats.n_chrom_runs;
    if (n_runs) {
        std::vector<uint64_t> first(n_runs);
        std::vector<char> names((size_t)n_runs * 32);
        if (n_runs <= HHGT_RESULT_RUNS) {
            memcpy(first.data(), hr->run_first, n_runs * 8);
            memcpy(names.data(), hr->run_names, (size_t)n_runs * 32);
        } else {
            HIP_TRY(hipMemcpy(first.data(), c->run_first.p, (size_t)n_runs * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(names.data(), c->run_names.p, (size_t)n_runs * 32, hipMemcpyDeviceToHost));
        }
        for (uint32_t r = 0; r < n_runs; ++r) {
            c->run_first_kept.push_back(first[r]);
            c->run_names_host.emplace_back(names.data() + (size_t)r * 32, strnlen(names.data() + (size_t)r * 32, 31));
        }
    }
    return HHGT_OK;
}

extern "C" int hhgt_encode_chrom_runs(hhgt_ctx *c, uint32_t max_runs, uint64_t *first_kept, char *names,
                                      uint32_t *n_runs)
{
    if (!c || !n_runs) return HHGT_ERR_ARG;
    uint32_t n = (uint32_t)c->run_first_kept.size();
    *n_runs = n;
    if (n > max_runs) n = max_runs;
    for (uint32_t r = 0; r < n; ++r) {
        if (first_kept) first_kept[r] = c->run_first_kept[r];
        if (names) {
            memset(names + (size_t)r * 32, 0, 32);
            size_t l = c->run_names_host[r].size() < 31 ? c->run_names_host[r].size() : 31;
            memcpy(names + (size_t)r * 32, c->run_names_host[r].data(), l);
        }
    }
    return HHGT_OK;
}

extern "C" int hhgt_pad_tail(hhgt_ctx *c, const hhgt_layout *lay, uint64_t v_end, uint64_t vcol_begin,
                             uint64_t vcol_end, void *d_G, void *stream)
{
    if (!c || !d_G) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    TRY(make_layout(lay, &L));
    if (v_end > L.v_capacity || vcol_end > L.v_capacity / L.Vc) {
        hhgt_set_error("pad_tail: range outside the layout");
        return HHGT_ERR_ARG;
    }
    return launch_pad_tail(L, v_end, vcol_begin, vcol_end, static_cast<int8_t *>(d_G),
                           reinterpret_cast<hipStream_t>(stream));
}

extern "C" int hhgt_pad_tail_cursor(hhgt_ctx *c, const hhgt_layout *lay, const uint64_t *d_cursor, void *d_G, void *stream)
{
    if (!c || !d_G || !d_cursor) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    LayoutDev L;
    TRY(make_layout(lay, &L));
    return launch_pad_tail_cursor(L, d_cursor, static_cast<int8_t *>(d_G), reinterpret_cast<hipStream_t>(stream));
}

// ---- compress -------------------------------------------------------------------------------------
static int check_codec_args(uint64_t chunk_nbytes, int typesize, int blocksize, int format)
{
    if (typesize < 1 || typesize > 255 || blocksize < 16 || blocksize > 65536 || blocksize % typesize ||
        chunk_nbytes == 0 || chunk_nbytes >= 0x7fffffffull || (format != HHGT_BLOSC1 && format != HHGT_BLOSC2)) {
        hhgt_set_error("compress: bad arguments (chunk_nbytes=%llu typesize=%d blocksize=%d format=%d)",
                       (unsigned long long)chunk_nbytes, typesize, blocksize, format);
        return HHGT_ERR_ARG;
    }
    return HHGT_OK;
}

// c-blosc never writes a blocksize larger than the chunk (its decoder rejects such headers)
static int effective_blocksize(uint64_t chunk_nbytes, int typesize, int blocksize)
{
    if ((uint64_t)blocksize > chunk_nbytes) {
        blocksize = (int)chunk_nbytes;
        if (typesize > 1 && blocksize >= typesize) blocksize -= blocksize % typesize;
    }
    return blocksize > 0 ? blocksize : 1;
}

static void codec_geometry(uint64_t chunk_nbytes, int typesize, int blocksize, uint32_t *nblocks, uint32_t *nwaves,
                           size_t *slot_bytes)
{
    const bool split = typesize >= 2 && typesize <= 16 && blocksize / typesize >= 128;
    *nblocks = (uint32_t)((chunk_nbytes + blocksize - 1) / blocksize);
    *nwaves = split ? (uint32_t)typesize : 1u;
    size_t max_stream = split ? (size_t)blocksize / typesize : (size_t)blocksize;
    if (split && chunk_nbytes % blocksize && chunk_nbytes % blocksize > max_stream) max_stream = chunk_nbytes % blocksize;
    *slot_bytes = lz4_slot_bytes((int)max_stream);
}

extern "C" uint64_t hhgt_compress_bound(uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize)
{
    if (check_codec_args(chunk_nbytes, typesize, blocksize, HHGT_BLOSC2) != HHGT_OK) return 0;
    // a chunk never exceeds nbytes + 32 (memcpyed form)
    return n_chunks * (chunk_nbytes + 32);
}

// shuffle + LZ4 + framing of n_chunks chunks that exist as int8 bytes (d_src) or as bit planes (d_planes; d_src then only
// holds the bytes of calls beyond 0 / 1 / missing and may be NULL)
static int compress_impl(hhgt_ctx *c, const void *d_src, const void *d_planes, PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes,
                         int typesize, int blocksize, int format, void *d_dst, uint64_t dst_cap,
                         uint64_t *d_chunk_off, uint64_t *total_bytes, void *stream)
{
    if (!c || (!d_src && !d_planes) || !d_dst || !d_chunk_off) return HHGT_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    TRY(check_codec_args(chunk_nbytes, typesize, blocksize, format));
    if (total_bytes) *total_bytes = 0;
    if (n_chunks == 0) return HHGT_OK;
    if (!total_bytes && dst_cap < n_chunks * (chunk_nbytes + 32)) {
        // without the read-back an overflow could not be reported: chunks that do not fit would silently be missing
        hhgt_set_error("compress: the asynchronous form (total_bytes == NULL) needs dst_cap >= hhgt_compress_bound() = %llu, got %llu",
                       (unsigned long long)(n_chunks * (chunk_nbytes + 32)), (unsigned long long)dst_cap);
        return HHGT_ERR_CAPACITY;
    }
    blocksize = effective_blocksize(chunk_nbytes, typesize, blocksize);
    uint32_t nblocks, nwaves;
    size_t slot;
    codec_geometry(chunk_nbytes, typesize, blocksize, &nblocks, &nwaves, &slot);
    const uint64_t n_streams = n_chunks * nblocks * nwaves;
    // With a frame stream the framing of this call runs there, behind the LZ4 kernels, while the caller's stream is free
    // for the LZ4 kernels of the next call — which write the other workspace set, and wait for the framing that last read it.
    hipStream_t fst = c->frame_stream ? c->frame_stream : st;
    hhgt_ctx::CodecWs &w = c->cw[c->frame_stream ? (c->cmp_seq++ & 1u) : 0u];
    TRY(w.lz_scratch.ensure((size_t)n_streams * slot));
    TRY(w.lz_csize.ensure((size_t)n_streams * 4));
    TRY(w.fr_bsize.ensure((size_t)n_chunks * nblocks * 4));
    TRY(w.fr_csize.ensure(((size_t)n_chunks + 1) * 8));
    TRY(w.fr_flags.ensure((size_t)n_chunks * 4));
    {   // k_frame_fused's state words carry the tag of their launch: zeroed when (re)allocated and when the tag wraps
        const size_t cap0 = w.fr_state.cap;
        TRY(w.fr_state.ensure(frame_state_bytes(n_chunks)));
        if (w.fr_state.cap != cap0 || ++w.fr_tag >= (1u << 20)) {
            HIP_TRY(hipMemsetAsync(w.fr_state.p, 0, w.fr_state.cap, fst));
            w.fr_tag = 1;
        }
    }
    if (w.fr_pending) {
        HIP_TRY(hipStreamWaitEvent(st, w.fr_done, 0));
        w.fr_pending = false;
    }
    {
        StageTimer t(c, st, HHGT_STAGE_LZ4);
        if (!w.lz_flags.p) {
            TRY(w.lz_flags.ensure(8));
            HIP_TRY(hipMemsetAsync(w.lz_flags.p, 0, 8, st));
        }
        if (++w.lz_tag == 0) w.lz_tag = 1;
        TRY(launch_lz4_blocks(static_cast<const uint8_t *>(d_src), static_cast<const uint8_t *>(d_planes), pg, n_chunks, chunk_nbytes, typesize,
                              blocksize, w.lz_scratch.as<uint8_t>(), slot, w.lz_csize.as<uint32_t>(), c->clevel, w.lz_flags.as<uint32_t>(),
                              w.lz_tag, st));
        t.stop();
    }
    if (fst != st) {
        if (!w.lz_done) HIP_TRY(hipEventCreateWithFlags(&w.lz_done, hipEventDisableTiming));
        if (!w.fr_done) HIP_TRY(hipEventCreateWithFlags(&w.fr_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(w.lz_done, st));
        HIP_TRY(hipStreamWaitEvent(fst, w.lz_done, 0));
    }
    {
        StageTimer t(c, fst, HHGT_STAGE_FRAME);
        TRY(launch_frame(w.lz_scratch.as<uint8_t>(), slot, w.lz_csize.as<uint32_t>(),
                         static_cast<const uint8_t *>(d_src), static_cast<const uint8_t *>(d_planes), pg, n_chunks, chunk_nbytes, typesize,
                         blocksize, format, w.fr_bsize.as<uint32_t>(), w.fr_csize.as<uint64_t>(), static_cast<uint8_t *>(d_dst),
                         dst_cap, d_chunk_off, w.fr_flags.as<uint32_t>(), w.fr_state.p, w.fr_tag, fst));
        t.stop();
    }
    if (fst != st) {
        HIP_TRY(hipEventRecord(w.fr_done, fst));
        w.fr_pending = true;
        if (total_bytes) HIP_TRY(hipStreamWaitEvent(st, w.fr_done, 0));   // the synchronous form: the caller's stream sees the chunks
    }
    if (total_bytes) {
        uint64_t tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, d_chunk_off + n_chunks, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *total_bytes = tot;
        if (tot > dst_cap) {
            hhgt_set_error("compress: need %llu bytes, dst_cap is %llu", (unsigned long long)tot,
                           (unsigned long long)dst_cap);
            return HHGT_ERR_CAPACITY;
        }
    }
    return HHGT_OK;
}

extern "C" int hhgt_compress_chunks(hhgt_ctx *c, const void *d_src, uint64_t n_chunks, uint64_t chunk_nbytes,
                                    int typesize, int blocksize, int format, void *d_dst, uint64_t dst_cap,
                                    uint64_t *d_chunk_off, uint64_t *total_bytes, void *stream)
{
    if (!d_src) return HHGT_ERR_ARG;
    return compress_impl(c, d_src, nullptr, PlanesGeom{0, 0, 0, 0}, n_chunks, chunk_nbytes, typesize, blocksize, format, d_dst, dst_cap,
                         d_chunk_off, total_bytes, stream);
}

extern "C" int hhgt_compress_planes(hhgt_ctx *c, const hhgt_layout *lay, const void *d_P, const void *d_G, uint32_t col0, uint32_t n_cols,
                                    int format, void *d_dst, uint64_t dst_cap, uint64_t *d_chunk_off, uint64_t *total_bytes,
                                    void *stream)
{
    if (!c || !d_P) return HHGT_ERR_ARG;
    LayoutDev L;
    PlanesGeom pg;
    TRY(planes_columns(lay, col0, n_cols, &L, &pg));
    const uint64_t chunk_nbytes = (uint64_t)L.Sc * L.Vc * 2ull, col_bytes = chunk_nbytes * L.n_sc;
    // the chunks of columns col0.. in the int8 matrix (it only backs the calls beyond 0 / 1 / missing)
    const uint8_t *g = d_G ? static_cast<const uint8_t *>(d_G) + (uint64_t)col0 * col_bytes : nullptr;
    return compress_impl(c, g, d_P, pg, (uint64_t)n_cols * L.n_sc, chunk_nbytes, 2, 8192, format, d_dst, dst_cap, d_chunk_off, total_bytes,
                         stream);
}

// Workspaces of the context made ahead of time for calls of a known size (they otherwise grow inside the first calls: hipMalloc
// of hundreds of MB each, in the middle of a pipeline's first pass).  text_bytes / max_lines: the largest encode call to come;
// n_chunks x chunk_nbytes (typesize, blocksize): the largest compress call.  Either half may be 0.
extern "C" int hhgt_reserve(hhgt_ctx *c, uint64_t text_bytes, uint32_t max_lines, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize,
                            int blocksize)
{
    if (!c) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (text_bytes) {
        const uint32_t n_regions = (uint32_t)((text_bytes + 1 + INDEX_REGION - 1) / INDEX_REGION);
        TRY(c->slots.ensure((size_t)n_regions * INDEX_CAP * 4));
        TRY(c->counts.ensure((size_t)n_regions * 4));
        TRY(c->prefix.ensure(((size_t)n_regions + 1) * 4));
        const size_t st_el = scan_tmp_elems(n_regions) > scan_tmp_elems(max_lines) ? scan_tmp_elems(n_regions) : scan_tmp_elems(max_lines);
        TRY(c->scan_tmp.ensure(2 * st_el * 4));
        const size_t nl4 = ((size_t)max_lines + 1) * 4;
        TRY(c->nl.ensure(nl4));
        DevBuf *per_line[] = {&c->l_soff, &c->l_lend, &c->l_pos, &c->l_refalt, &c->l_flags, &c->l_keep, &c->l_kidx, &c->l_cnew, &c->l_crun,
                              &c->k_soff, &c->k_lend, &c->k_meta, &c->redo_list, &c->redo_flag};
        for (DevBuf *b : per_line) TRY(b->ensure(nl4));
        TRY(c->run_first.ensure(MAX_CHROM_RUNS * 8));
        TRY(c->run_names.ensure(MAX_CHROM_RUNS * 32));
        TRY(c->result.ensure(sizeof(hhgt_encode_result)));
    }
    if (n_chunks) {
        TRY(check_codec_args(chunk_nbytes, typesize, blocksize, HHGT_BLOSC2));
        blocksize = effective_blocksize(chunk_nbytes, typesize, blocksize);
        uint32_t nblocks, nwaves;
        size_t slot;
        codec_geometry(chunk_nbytes, typesize, blocksize, &nblocks, &nwaves, &slot);
        const uint64_t n_streams = n_chunks * nblocks * nwaves;
        hhgt_ctx::CodecWs &w = c->cw[0];
        TRY(w.lz_scratch.ensure((size_t)n_streams * slot));
        TRY(w.lz_csize.ensure((size_t)n_streams * 4));
        TRY(w.fr_bsize.ensure((size_t)n_chunks * nblocks * 4));
        TRY(w.fr_csize.ensure(((size_t)n_chunks + 1) * 8));
        TRY(w.fr_flags.ensure((size_t)n_chunks * 4));
        const size_t cap0 = w.fr_state.cap;
        TRY(w.fr_state.ensure(frame_state_bytes(n_chunks)));
        if (w.fr_state.cap != cap0) {
            HIP_TRY(hipMemset(w.fr_state.p, 0, w.fr_state.cap));
            w.fr_tag = 0;
        }
    }
    return HHGT_OK;
}

extern "C" int hhgt_set_clevel(hhgt_ctx *c, int clevel)
{
    if (!c || clevel < 1 || clevel > 9) {
        hhgt_set_error("clevel must be 1..9");
        return HHGT_ERR_ARG;
    }
    c->clevel = clevel;
    return HHGT_OK;
}

extern "C" int hhgt_stream_create(hhgt_ctx *c, int kind, void **out)
{
    if (!c || !out || (kind != HHGT_STREAM_ENCODE && kind != HHGT_STREAM_COMPRESS && kind != HHGT_STREAM_FRAME)) {
        hhgt_set_error("hhgt_stream_create: bad arguments");
        return HHGT_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = nullptr;
    if (kind == HHGT_STREAM_FRAME) {
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    } else if (kind == HHGT_STREAM_ENCODE) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    } else {
        const uint32_t n_cu = (uint32_t)c->prop.multiProcessorCount;
        // the mask works in groups of 32 CUs on this part (one XCD): 3/4 of the chip, rounded to that
        uint32_t use = n_cu >= 128u ? (n_cu * 3u / 4u) / 32u * 32u : n_cu;
        if (const char *e = getenv("HHGT_COMPRESS_CUS")) {
            const int v = atoi(e);
            use = v <= 0 || (uint32_t)v >= n_cu ? n_cu : (uint32_t)v;
        }
        if (use >= n_cu) {
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        } else {
            std::vector<uint32_t> mask((n_cu + 31u) / 32u, 0u);
            for (uint32_t i = n_cu - use; i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31u);   // the last `use` CUs of the mask order
            if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();   // a runtime that refuses the mask: an ordinary stream does the same work
                st = nullptr;
                HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            }
        }
    }
    *out = st;
    return HHGT_OK;
}

extern "C" int hhgt_set_frame_stream(hhgt_ctx *c, void *stream)
{
    if (!c) return HHGT_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    // framings still in flight on the old stream read the workspace sets: let them finish before the roles change
    if (c->frame_stream) HIP_TRY(hipStreamSynchronize(c->frame_stream));
    for (auto &w : c->cw) w.fr_pending = false;
    c->frame_stream = reinterpret_cast<hipStream_t>(stream);
    return HHGT_OK;
}

extern "C" int hhgt_stream_destroy(hhgt_ctx *c, void *stream)
{
    if (!c) return HHGT_ERR_ARG;
    if (stream) {
        HIP_TRY(hipSetDevice(c->device));
        if (reinterpret_cast<hipStream_t>(stream) == c->frame_stream) TRY(hhgt_set_frame_stream(c, nullptr));
        HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    }
    return HHGT_OK;
}

extern "C" int hhgt_set_index_mode(hhgt_ctx *c, int mode)
{
    if (!c || mode < -1 || mode > 2) {
        hhgt_set_error("index mode must be -1 (default), 0 (scan), 1 (hop) or 2 (walk)");
        return HHGT_ERR_ARG;
    }
    c->index_mode = mode;
    return HHGT_OK;
}

extern "C" int hhgt_set_keep_multiallelic(hhgt_ctx *c, int on)
{
    if (!c) return HHGT_ERR_ARG;
    c->keep_multi = on ? 1 : 0;
    return HHGT_OK;
}

__global__ void k_count_nonzero_u32(const uint32_t *__restrict__ v, uint64_t n, unsigned long long *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && v[i] != 0u);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(out, (unsigned long long)__builtin_popcountll(m));
}

extern "C" int hhgt_inflate_members(hhgt_ctx *c, const void *d_src, uint64_t src_bytes, const uint64_t *d_comp_off,
                                    const uint32_t *d_comp_len, const uint64_t *d_out_off, const uint32_t *d_isize,
                                    uint64_t n_members, void *d_dst, uint64_t dst_bytes, const uint32_t *d_crc32,
                                    uint32_t *d_status, uint64_t *n_bad, void *stream)
{
    if (!c || !d_src || !d_comp_off || !d_comp_len || !d_out_off || !d_isize || !d_status || (!d_dst && dst_bytes))
        return HHGT_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    if (n_bad) *n_bad = 0;
    if (n_members == 0) return HHGT_OK;
    if (d_crc32 && !c->crc_x2n_ready) {
        uint32_t t[32];
        crc32_x2n_table(t);
        TRY(c->crc_x2n.ensure(sizeof(t)));
        HIP_TRY(hipMemcpy(c->crc_x2n.p, t, sizeof(t), hipMemcpyHostToDevice));
        c->crc_x2n_ready = true;
    }
    {
        StageTimer t(c, st, HHGT_STAGE_INFLATE);
        TRY(launch_inflate(static_cast<const uint8_t *>(d_src), src_bytes, d_comp_off, d_comp_len, d_out_off, d_isize,
                           n_members, static_cast<uint8_t *>(d_dst), dst_bytes, d_status, d_crc32,
                           c->crc_x2n.as<uint32_t>(), st));
        t.stop();
    }
    if (n_bad) {
        TRY(c->dec_bad.ensure(8));
        HIP_TRY(hipMemsetAsync(c->dec_bad.p, 0, 8, st));
        hipLaunchKernelGGL(k_count_nonzero_u32, dim3((uint32_t)((n_members + 255) / 256)), dim3(256), 0, st, d_status,
                           n_members, c->dec_bad.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        uint64_t nb = 0;
        HIP_TRY(hipMemcpyAsync(&nb, c->dec_bad.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *n_bad = nb;
    }
    return HHGT_OK;
}

extern "C" int hhgt_decompress_chunks(hhgt_ctx *c, const void *d_src, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                      uint64_t chunk_nbytes, int typesize, int blocksize, void *d_dst,
                                      uint64_t *n_bad, void *stream)
{
    if (!c || !d_src || !d_dst || !d_chunk_off) return HHGT_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    TRY(check_codec_args(chunk_nbytes, typesize, blocksize, HHGT_BLOSC2));
    if (n_bad) *n_bad = 0;
    if (n_chunks == 0) return HHGT_OK;
    blocksize = effective_blocksize(chunk_nbytes, typesize, blocksize);
    TRY(c->dec_bad.ensure(8));
    HIP_TRY(hipMemsetAsync(c->dec_bad.p, 0, 8, st));
    {
        StageTimer t(c, st, HHGT_STAGE_DECODE);
        TRY(launch_decode(static_cast<const uint8_t *>(d_src), d_chunk_off, n_chunks, chunk_nbytes, typesize,
                          blocksize, static_cast<uint8_t *>(d_dst), c->dec_bad.as<unsigned long long>(), st));
        t.stop();
    }
    if (n_bad) {
        uint64_t nb = 0;
        HIP_TRY(hipMemcpyAsync(&nb, c->dec_bad.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *n_bad = nb;
    }
    return HHGT_OK;
}
