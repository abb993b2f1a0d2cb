// Microbenchmark: cost of unaligned LDS reads on gfx950 (ds_read_b32 / b64 / b128 at byte addresses).
// build: hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip ; run: ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct __attribute__((packed)) P4 { uint32_t a; };
struct __attribute__((packed)) P8 { uint32_t a, b; };
struct __attribute__((packed)) P16 { uint32_t a, b, c, d; };
template <int BYTES, int STRIDE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t iters, uint32_t misalign)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768 + 64];
    for (uint32_t i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t lane_addr = (threadIdx.x & 63u) * STRIDE + (threadIdx.x >> 6) * 4096u + misalign;
#pragma unroll 8
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t addr = lane_addr + ((it * 272u) & 0x3ff0u);
        if constexpr (BYTES == 4) acc += reinterpret_cast<const P4 *>(lds + addr)->a;
        else if constexpr (BYTES == 8) { P8 v = *reinterpret_cast<const P8 *>(lds + addr); acc += v.a ^ v.b; }
        else { P16 v = *reinterpret_cast<const P16 *>(lds + addr); acc += v.a ^ v.b ^ v.c ^ v.d; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int BYTES, int STRIDE>
static void run(const char *name, uint32_t mis)
{
    uint32_t *d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t iters = 4096;
    k<BYTES, STRIDE><<<1024, 256>>>(d, 16, mis);
    hipEventRecord(a);
    k<BYTES, STRIDE><<<1024, 256>>>(d, iters, mis);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts = 1024.0 * 4 * iters;  // wave-instructions
    printf("%-28s misalign %u: %.3f ms, %.2f ns per wave-instruction per CU-slot (256 CUs): %.1f cycles@2.1GHz/CU\n", name, mis, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 256 * 2.1);
    hipFree(d);
}
int main()
{
    for (uint32_t mis : {0u, 1u, 2u}) {
        run<4, 4>("b32 lane stride 4", mis);
        run<4, 1>("b32 lane stride 1", mis);
        run<8, 8>("b64 lane stride 8", mis);
        run<16, 16>("b128 lane stride 16", mis);
        run<16, 1>("b128 lane stride 1", mis);
    }
    return 0;
}
