// In which order does the LDS serve the lanes of ONE ds_wrxchg_rtn_b32 that hit the same address?  If it is ascending
// lane order, atomicExch(tab[key], lane) hands every lane the previous lane with the same key — the sequential
// "most recent earlier occurrence" semantics of a hash table, for 64 insertions in one instruction.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xo tools/micro/lds_xchg_order.hip && /tmp/xo
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out, int nkeys, unsigned seed)
{
    __shared__ unsigned tab[64];
    const unsigned lane = threadIdx.x;
    if (lane < 64) tab[lane] = 0xFFFFu;
    __syncthreads();
    unsigned key = (lane * 2654435761u + seed) >> 7;
    key = (key ^ (key >> 5)) % (unsigned)nkeys;
    const unsigned prev = atomicExch(&tab[key], lane);
    out[blockIdx.x * 128 + lane] = prev;
    out[blockIdx.x * 128 + 64 + lane] = key;
}
int main()
{
    unsigned *d, h[128 * 64];
    hipMalloc(&d, sizeof(h));
    int bad = 0, total = 0;
    for (int nk = 1; nk <= 64; nk *= 2) {
        hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d, nk, 12345u * nk);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 64; ++b)
            for (int l = 0; l < 64; ++l) {
                unsigned key = h[b * 128 + 64 + l], want = 0xFFFFu;
                for (int j = l - 1; j >= 0; --j)
                    if (h[b * 128 + 64 + j] == key) { want = j; break; }
                ++total;
                if (h[b * 128 + l] != want) ++bad;
            }
    }
    printf("lanes checked %d, not 'previous lane with the same key': %d\n", total, bad);
    return 0;
}
