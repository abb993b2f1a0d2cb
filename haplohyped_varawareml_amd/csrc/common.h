// common.h — internal declarations shared by the libhhgt.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/hhgt.h"

#define HHGT_WAVE 64
// development (-DHHGT_ENC_PRIO=n): the memory-bound kernels of the encode chain raise their waves' issue priority, so that
// beside the (VALU-bound) LZ4 kernel of the other stream their few instructions go out first
#ifdef HHGT_ENC_PRIO
#define HHGT_WAVE_PRIO() __builtin_amdgcn_s_setprio(HHGT_ENC_PRIO)
#else
#define HHGT_WAVE_PRIO()
#endif

// wave-wide inclusive sum on DPP row shifts (no LDS round trips)
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t wave_scan_sum_dpp(uint32_t x, uint32_t lane)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31) + t0;
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + t1;
    const uint32_t row = lane >> 4;
    return x + (row == 0u ? 0u : (row == 1u ? t0 : (row == 2u ? t1 : t2)));
}
#endif

// ---- error plumbing -------------------------------------------------------------------------
void hhgt_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            hhgt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                      \
            return HHGT_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

// ---- per-line flag bits (index/fixed-column stage) --------------------------------------------
enum : uint32_t {
    LF_RECORD = 1u,        // data line (non-empty, not '#')
    LF_KEEP = 2u,          // passed region + isSNP
    LF_FAST = 4u,          // FORMAT == "GT" and sample region is exactly 4*S-1 bytes
    LF_MALFORMED = 8u,
    LF_DROP_REGION = 16u,
    LF_DROP_FILTER = 32u,
    LF_CHROM_NEW = 64u,    // CHROM differs from the previous data line
};

// device-side counters written by the kernels of one hhgt_encode_text call
struct DevCounters {
    unsigned long long n_lines;
    unsigned long long n_records;
    unsigned long long n_kept;
    unsigned long long n_drop_region;
    unsigned long long n_drop_filter;
    unsigned long long n_haploid;
    unsigned long long n_malformed;
    unsigned long long n_general;    // entries in the redo (variable-width) list
    unsigned long long n_chrom_runs;
    unsigned long long err_density;  // newline-slot overflow
    unsigned long long err_lines;    // lines beyond the caller's max_lines (asynchronous form only)
    unsigned long long cursor_after; // *d_cursor after the call
    unsigned long long n_other;      // bit-plane form: calls that are neither 0, 1 nor missing (their bytes went to d_G)
    unsigned long long pad[3];
};

#define MAX_CHROM_RUNS 4096u

struct RegionFilter {
    char contig[64];
    int contig_len;   // 0 = no filter
    int has_range;
    long long beg, end;  // 1-based inclusive
    int keep_multi;      // non-reference mode: multi-allelic SNP sites pass the record filter (hhgt_set_keep_multiallelic)
};

// index stage geometry: one wave scans INDEX_REGION bytes, at most INDEX_CAP newlines in it
#define INDEX_REGION 16384u
#define INDEX_CAP 1024u

// encode tile geometry
#define TILE_V 128
#define TILE_S 256

struct LayoutDev {
    uint32_t S;
    uint32_t sc_log2;    // log2(samples per chunk); 31 = dense (single sample-chunk)
    uint32_t n_sc;       // sample-chunks per chunk column
    uint32_t Sc;         // samples per chunk (dense: S)
    uint64_t Vc;         // variants per chunk (dense: v_capacity)
    uint64_t v_capacity;
    uint32_t ring;       // > 0: G is a ring of `ring` chunk columns (v_capacity = ring * Vc), kept indices unbounded
    uint32_t pad_;
};

// ---- bit-plane form of the matrix (include/hhgt.h "Bit-plane form"): tile-major ----------------------------------------
// A chunk column (Vc variants x S_pad padded sample rows) is cut into tiles of PL_TILE variants; a tile holds, per kind-plane
// kp (0: ONE hap 0, 1: ONE hap 1, 2: EXC hap 0, 3: EXC hap 1) and sample row r, one 32-byte piece (bit i = variant i of the
// tile): P[column slot][tile][kp][row][32 B].  The encoder's workgroup (256 rows x one tile) therefore writes four
// contiguous 8 KiB runs — the first version of this path wrote row-major planes, 32-byte pieces 2 KiB apart, and spent
// 40 % of its time on those partial-line writes — and a compressor wave gathers the 16 pieces of its 4096-variant plane.
#define PL_TILE 256u
struct PlanesGeom {
    uint32_t S_pad;   // padded sample rows per chunk column (n_sc * Sc)
    uint32_t bpr;     // Blosc blocks (4096 variants) per sample row of a column: Vc / 4096
    uint32_t tpc;     // tiles per column: Vc / PL_TILE = 16 * bpr
    uint32_t col0;    // first column slot a compress / expand call works on
};
static inline __host__ __device__ uint64_t planes_piece(const PlanesGeom &g, uint64_t col, uint32_t tile, uint32_t kp, uint32_t row)
{
    return ((((col * g.tpc + tile) * 4ull + kp) * g.S_pad) + row) * 32ull;
}
static inline __host__ __device__ PlanesGeom planes_geom(const LayoutDev &L, uint32_t col0)
{
    PlanesGeom g;
    g.S_pad = L.n_sc * L.Sc;
    g.bpr = (uint32_t)(L.Vc >> 12);
    g.tpc = (uint32_t)(L.Vc / PL_TILE);
    g.col0 = col0;
    return g;
}
// block `id` of a compress call (G's block order relative to column col0) -> column slot, sample row, block in the row
static inline __host__ __device__ void planes_block(const PlanesGeom &g, uint64_t id, uint64_t *col, uint32_t *row, uint32_t *bi)
{
    const uint64_t per_col = (uint64_t)g.S_pad * g.bpr;
    const uint64_t c = id / per_col;
    const uint32_t rem = (uint32_t)(id - c * per_col);
    *col = g.col0 + c;
    *row = rem / g.bpr;
    *bi = rem - (rem / g.bpr) * g.bpr;
}

// workspace buffer that only grows
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct hhgt_ctx {
    int device = 0;
    hipDeviceProp_t prop;
    // index / fixed / keep workspaces
    DevBuf slots, counts, prefix, nl, scan_tmp;
    DevBuf l_soff, l_lend, l_pos, l_refalt, l_flags, l_keep, l_kidx, l_cnew, l_crun;
    DevBuf k_soff, k_lend, k_meta, redo_list, redo_flag, run_first, run_names;
    DevBuf counters;       // DevCounters
    DevBuf cursor;         // uint64: v_base of the synchronous hhgt_encode_text (the asynchronous form gets the caller's)
    DevBuf result;         // hhgt_encode_result staging of the asynchronous form
    // compress workspaces
    // two sets: with a frame stream (hhgt_set_frame_stream) the framing of call k reads set k & 1 while the LZ4 kernels of
    // call k + 1 write the other
    struct CodecWs {
        DevBuf lz_scratch, lz_csize, lz_flags, fr_bsize, fr_csize, fr_flags, fr_state;
        uint32_t lz_tag = 0;       // tag of the last launch_lz4_blocks on lz_flags
        uint32_t fr_tag = 0;       // launch tag of k_frame_fused's state words (1 .. 2^20 - 1, then the buffer is zeroed again)
        hipEvent_t lz_done = nullptr, fr_done = nullptr;
        bool fr_pending = false;   // fr_done was recorded behind a framing that read this set
    } cw[2];
    uint64_t cmp_seq = 0;
    hipStream_t frame_stream = nullptr;   // hhgt_set_frame_stream; nullptr: the framing follows the LZ4 kernels on their stream
    DevBuf dec_bad, oh_ovl, oh_lut, crc_x2n;
    bool crc_x2n_ready = false;
    // pinned host mirror for counters
    DevCounters *h_counters = nullptr;
    hhgt_encode_result *h_result_pinned = nullptr;
    hhgt_encode_result h_result;
    // last encode's chrom runs (host)
    std::vector<uint64_t> run_first_kept;
    std::vector<std::string> run_names_host;
    int clevel = 5;        // Blosc clevel analogue (reference: compression_opts[4] = 5)
    int keep_multi = 0;    // hhgt_set_keep_multiallelic
    int index_mode = -1;   // hhgt_set_index_mode (< 0: HHGT_INDEX_MODE, default: the walk)
    bool counters_clean = false;   // k_encode_finish of the last call left the DevCounters zeroed
    // profiling
    int profiling = 0;
    double stage_ms[HHGT_N_STAGES] = {0};
    uint64_t stage_launches[HHGT_N_STAGES] = {0};
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
};

// RAII-ish stage timer: records events around a stage when profiling is on
struct StageTimer {
    hhgt_ctx *ctx;
    hipStream_t st;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(hhgt_ctx *c, hipStream_t s, int stage_);
    void stop();
};
int hhgt_profile_collect(hhgt_ctx *ctx);

// ---- launchers (defined in the .hip files) ------------------------------------------------------
// scan.hip
int launch_scan_exclusive_u32(const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp,
                              size_t tmp_elems, hipStream_t st);
int launch_scan_exclusive_u32_pair(const uint32_t *in_a, uint32_t *out_a, const uint32_t *in_b, uint32_t *out_b, uint64_t n,
                                   uint32_t *d_tmp, size_t tmp_elems, hipStream_t st);
size_t scan_tmp_elems(uint64_t n);

// index.hip
int launch_index_newlines(const uint8_t *d_text, uint64_t n, uint32_t *d_slots, uint32_t *d_counts,
                          uint32_t n_regions, uint32_t min_line, uint32_t S, int mode, DevCounters *d_cnt, hipStream_t st);
int index_mode_default();
// Everything after the newline index is sized by a host-side BOUND on the line count (max_lines) and reads the
// actual count (d_nlines = prefix[n_regions]) and the append position (d_cursor) from device memory, so the chain
// can be queued without a host round trip (hhgt_encode_text_async).
int launch_compact_newlines(const uint32_t *d_slots, const uint32_t *d_counts, const uint32_t *d_prefix,
                            uint32_t n_regions, uint32_t *d_nl, uint32_t max_lines, hipStream_t st);
int launch_parse_fixed(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl, const uint32_t *d_nlines,
                       uint32_t max_lines, const RegionFilter &region, uint32_t S, uint32_t *l_soff, uint32_t *l_lend,
                       uint32_t *l_pos, uint32_t *l_refalt, uint32_t *l_flags, uint32_t *l_keep,
                       uint32_t *l_cnew, int mode, DevCounters *d_cnt, hipStream_t st);
int launch_compact_kept(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl, const uint32_t *d_nlines, uint32_t max_lines,
                        const uint32_t *l_soff, const uint32_t *l_lend,
                        const uint32_t *l_pos, const uint32_t *l_refalt, const uint32_t *l_flags,
                        const uint32_t *l_kidx, const uint32_t *l_crun, uint32_t *k_soff, uint32_t *k_lend,
                        uint32_t *k_meta, uint32_t *redo_list, uint32_t *redo_flag, uint64_t *run_first, uint8_t *run_names,
                        uint32_t max_runs, const uint64_t *d_cursor, uint64_t v_capacity, uint32_t ring, uint32_t *d_start,
                        uint32_t *d_stop, uint8_t *d_ref, uint8_t *d_alt, DevCounters *d_cnt, bool strided, hipStream_t st);

// encode.hip
int launch_encode_tiles(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_meta,
                        uint32_t n_lines_bound, const uint64_t *d_cursor, LayoutDev lay, int8_t *d_G,
                        uint32_t *redo_list, uint32_t *redo_flag, DevCounters *d_cnt, hipStream_t st);
int launch_encode_general(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_lend,
                          const uint32_t *k_meta, const uint32_t *redo_list, const uint64_t *d_cursor, LayoutDev lay,
                          int8_t *d_G, uint8_t *d_P, DevCounters *d_cnt, int n_cu, hipStream_t st);
// bit-plane form of the matrix (include/hhgt.h): 2048 plane bytes per 8 KiB block of G
int launch_encode_planes(const uint8_t *d_text, uint64_t n, const uint32_t *k_soff, const uint32_t *k_meta,
                         uint32_t n_lines_bound, const uint64_t *d_cursor, LayoutDev lay, uint8_t *d_P, int8_t *d_G,
                         uint32_t *redo_list, uint32_t *redo_flag, DevCounters *d_cnt, hipStream_t st);
int launch_pad_tail_planes(LayoutDev lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end, uint8_t *d_P, hipStream_t st);
int launch_pad_tail_planes_cursor(LayoutDev lay, const uint64_t *d_cursor, uint8_t *d_P, hipStream_t st);
int launch_planes_expand(LayoutDev lay, const uint8_t *d_P, const uint8_t *d_G, uint32_t col0, uint32_t n_cols, uint8_t *d_out, hipStream_t st);
int launch_pad_tail(LayoutDev lay, uint64_t v_end, uint64_t vcol_begin, uint64_t vcol_end, int8_t *d_G,
                    hipStream_t st);
// zero [*d_cursor, round_up(*d_cursor, Vc)) of the cursor's chunk column and the sample padding rows of that column
int launch_pad_tail_cursor(LayoutDev lay, const uint64_t *d_cursor, int8_t *d_G, hipStream_t st);

// lz4.hip
size_t lz4_slot_bytes(int neblock);
// d_planes != NULL: the chunks exist as bit planes (hhgt.h "Bit-plane form"); d_src then only supplies the bytes of calls
// beyond 0 / 1 / missing and may be NULL
int launch_lz4_blocks(const uint8_t *d_src, const uint8_t *d_planes, PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize,
                      int blocksize, uint8_t *d_scratch, size_t slot_bytes, uint32_t *d_csize, int clevel, uint32_t *d_flags, uint32_t tag,
                      hipStream_t st);
// d_flags (two words, zero once; may be NULL) + tag (a value no earlier call on these words used, not 0): the coders note in
// them whether they left streams marked, and the scanning launches behind return at once when nobody did
// lz4bits.hip: typesize 2, 8 KiB blocks; streams it cannot code get csize = 0xFFFFFFFF
int launch_lz4_bitplanes(const uint8_t *d_src, bool planes, PlanesGeom pg, uint64_t n_blocks, uint8_t *d_scratch, size_t slot_bytes, uint32_t *d_csize,
                         int depth, uint32_t *d_flags, uint32_t tag, bool *exc_ran, hipStream_t st);
// frame.hip
int launch_frame(const uint8_t *d_scratch, size_t slot_bytes, const uint32_t *d_csize, const uint8_t *d_src, const uint8_t *d_planes,
                 PlanesGeom pg, uint64_t n_chunks, uint64_t chunk_nbytes, int typesize, int blocksize, int format,
                 uint32_t *d_bstart, uint64_t *d_chunk_csize, uint8_t *d_dst, uint64_t dst_cap,
                 uint64_t *d_chunk_off, uint32_t *d_chunk_flags, void *d_state, uint32_t tag, hipStream_t st);
size_t frame_state_bytes(uint64_t n_chunks);   // d_state: zeroed when allocated, then only ever written by the kernel (tagged words)
int launch_decode(const uint8_t *d_src, const uint64_t *d_chunk_off, uint64_t n_chunks, uint64_t chunk_nbytes,
                  int typesize, int blocksize, uint8_t *d_dst, unsigned long long *d_bad, hipStream_t st);
int launch_inflate(const uint8_t *d_src, uint64_t src_bytes, const uint64_t *d_comp_off, const uint32_t *d_comp_len,
                   const uint64_t *d_out_off, const uint32_t *d_isize, uint64_t n_members, uint8_t *d_dst,
                   uint64_t dst_bytes, uint32_t *d_status, const uint32_t *d_crc32, const uint32_t *d_x2n, hipStream_t st);
void crc32_x2n_table(uint32_t *t /*[32]*/);

// layout helper shared by host and device
static inline __host__ __device__ uint64_t layout_offset(const LayoutDev &L, uint32_t s, uint64_t v)
{
    uint64_t vcol = v / L.Vc, vin = v - vcol * L.Vc;
    if (L.ring) vcol %= L.ring;
    uint32_t scol = (L.sc_log2 >= 31) ? 0u : (s >> L.sc_log2);
    uint32_t sin = s - scol * L.Sc;
    return (((vcol * L.n_sc + scol) * L.Sc + sin) * L.Vc + vin) * 2ull;
}
