// Microbenchmark: what HBM gives for the encode kernels' read pattern — a workgroup owns a band of sample columns
// (PIECE bytes of every line) of TV consecutive lines of a text whose lines are ~10 KB apart — against tile shape, loads
// in flight per wave, waves per CU, load flavour and tile order.
// build: hipcc -O3 --offload-arch=gfx950 -o strided_read strided_read.hip ; run: ./strided_read
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4_un __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4_al __attribute__((ext_vector_type(4)));

// NW waves per workgroup, each wave reads LW lines, G loads in flight, lane reads 16 bytes of each line (a wave: 1 KiB)
// NT: nontemporal loads; SFAST: blockIdx.x walks the sample bands first (the bands of a line are read together)
template <int NW, int LW, int G, bool NT, bool SFAST>
__global__ __launch_bounds__(64 * NW) void k_read(const uint8_t *__restrict__ text, uint64_t stride, uint32_t n_lines, uint32_t n_bands,
                                                  uint32_t misalign, uint32_t *__restrict__ out)
{
    extern __shared__ uint8_t pad[];   // dynamic LDS only limits the workgroups per CU
    const uint32_t tiles_v = (n_lines + NW * LW - 1) / (NW * LW);
    uint32_t tv, band;
    if (SFAST) {
        band = blockIdx.x % n_bands;
        tv = blockIdx.x / n_bands;
    } else {
        tv = blockIdx.x % tiles_v;
        band = blockIdx.x / tiles_v;
    }
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t l0 = (tv * NW + w) * LW;
    uint32_t acc = 0;
    for (int g = 0; g < LW; g += G) {
        u32x4_al v[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t line = l0 + g + j;
            const uint8_t *p = text + (uint64_t)(line < n_lines ? line : 0u) * stride + misalign + band * 1024u + lane * 16u;
            if (NT) {
                u32x4_un t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_un *>(p));
                v[j] = u32x4_al{t.x, t.y, t.z, t.w};
            } else {
                u32x4_un t = *reinterpret_cast<const u32x4_un *>(p);
                v[j] = u32x4_al{t.x, t.y, t.z, t.w};
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc + pad[0];
}

template <int NW, int LW, int G, bool NT, bool SFAST>
static void run(const uint8_t *d, uint64_t stride, uint32_t n_lines, uint32_t n_bands, uint32_t mis, uint32_t *out, uint32_t wg_per_cu)
{
    const uint32_t tiles_v = (n_lines + NW * LW - 1) / (NW * LW);
    const size_t lds = wg_per_cu ? (160 * 1024) / wg_per_cu - 512 : 0;
    if (lds > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_read<NW, LW, G, NT, SFAST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_read<NW, LW, G, NT, SFAST>), dim3(tiles_v * n_bands), dim3(64 * NW), lds, 0, d, stride, n_lines, n_bands, mis, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)n_lines * n_bands * 1024.0;
    printf("waves/wg %d lines/wave %3d in-flight %2d %s %s wg/cu %2u misalign %2u: %.3f ms  %.2f TB/s\n", NW, LW, G, NT ? "nt" : "  ",
           SFAST ? "band-fastest" : "line-fastest", wg_per_cu, mis, best, bytes / best / 1e9);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main(int argc, char **argv)
{
    const uint32_t S = 2504, n_lines = 200000, n_bands = 9;   // 9 whole 1 KiB bands of the 10016 sample bytes
    const uint64_t stride = 4ull * S + (argc > 1 ? (uint64_t)atoi(argv[1]) : 58);
    uint8_t *d;
    uint32_t *out;
    if (hipMalloc(&d, stride * n_lines + 65536) != hipSuccess) return 1;
    hipMemset(d, 0x30, stride * n_lines + 65536);
    hipMalloc(&out, 1 << 22);
    printf("text %.2f GB, lines of %llu bytes; bytes read per run %.2f GB\n", stride * n_lines / 1e9, (unsigned long long)stride,
           n_lines * n_bands * 1024.0 / 1e9);
    if (argc > 2) {   // alignment sweep only: ./strided_read <fixed bytes per line> x
        for (uint32_t mis : {0u, 16u, 8u, 4u, 2u, 1u, 3u, 5u, 7u, 13u})
            run<4, 16, 16, true, false>(d, stride, n_lines, n_bands, mis, out, 4);
        return 0;
    }
    for (uint32_t mis : {56u, 64u}) {
        for (uint32_t wpc : {2u, 4u, 8u}) {
            run<4, 128, 8, true, false>(d, stride, n_lines, n_bands, mis, out, wpc);
            run<4, 128, 16, true, false>(d, stride, n_lines, n_bands, mis, out, wpc);
            run<4, 128, 32, true, false>(d, stride, n_lines, n_bands, mis, out, wpc);
            run<4, 16, 16, true, false>(d, stride, n_lines, n_bands, mis, out, wpc);
        }
        run<4, 128, 16, false, false>(d, stride, n_lines, n_bands, mis, out, 4);
        run<4, 128, 16, true, true>(d, stride, n_lines, n_bands, mis, out, 4);
        run<4, 128, 16, false, true>(d, stride, n_lines, n_bands, mis, out, 4);
        run<4, 16, 16, true, true>(d, stride, n_lines, n_bands, mis, out, 8);
        run<8, 64, 16, true, false>(d, stride, n_lines, n_bands, mis, out, 2);
        run<8, 64, 16, true, true>(d, stride, n_lines, n_bands, mis, out, 2);
        run<16, 32, 16, true, false>(d, stride, n_lines, n_bands, mis, out, 1);
    }
    // the same bytes as one flat stream (lines of exactly n_bands KiB): the ceiling
    run<4, 128, 16, true, true>(d, 1024ull * n_bands, n_lines, n_bands, 0, out, 4);
    run<4, 128, 16, true, false>(d, 1024ull * n_bands, n_lines, n_bands, 0, out, 4);
    hipFree(d);
    hipFree(out);
    return 0;
}
