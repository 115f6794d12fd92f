// Microbenchmark (round 4): returning LDS atomic OR (the test-and-set of the symbolic walk) when K lanes of a wave
// instruction hit the SAME 32-bit word with different bits -- operands with dense runs of columns (bands, blocks) do that:
// 128 consecutive columns are 4 bitmap words.
//   hipcc --offload-arch=gfx950 -O3 lds_or_sameword.hip -o lds_or_sameword && ./lds_or_sameword
// 16 waves per workgroup, one workgroup per CU, each wave on a bitmap of its own (1600 words); per instruction lane l
// targets word base + l / K (K = 1: 64 distinct words ... K = 64: one word), bit l % 32.
// Variants: atomicOr returning (what the walk uses); ds_read + non-returning ds_or; ds_read only (floor).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K, int VAR>
__global__ __launch_bounds__(1024) void k(int iters, unsigned *out)
{
    extern __shared__ unsigned bm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *my = bm + wave * 1664;
    for (int x = lane; x < 1664; x += 64) my[x] = 0u;
    __syncthreads();
    unsigned acc = 0, s = threadIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            s = s * 1664525u + 1013904223u;
            const int base = (__builtin_amdgcn_readfirstlane(s) >> 8) % 1536;       // wave-uniform start word
            const int w = base + lane / K;
            const unsigned bit = 1u << (lane & 31);
            if (VAR == 0) acc += atomicOr(my + w, bit);
            else if (VAR == 1) { acc += my[w]; __hip_atomic_fetch_or(my + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else acc += ((volatile unsigned *)my)[w];
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int K, int VAR>
void run(const char *name)
{
    const int grid = 256, iters = 2000;
    unsigned *o; hipMalloc(&o, (size_t)grid * 1024 * 4);
    const size_t lds = 16 * 1664 * 4;
    hipFuncSetAttribute((const void *)k<K, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K, VAR><<<grid, 1024, lds>>>(10, o);
    hipEventRecord(e0);
    k<K, VAR><<<grid, 1024, lds>>>(iters, o);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s K=%2d %8.3f ms  %6.1f cycles per wave instruction and CU at 2.1 GHz  [%s]\n", name, K, ms,
           ms * 1e-3 * 2.1e9 / (16.0 * iters * 16.0), hipGetErrorString(hipGetLastError()));
    hipFree(o);
}

int main()
{
#define ROW(K) run<K, 0>("atomicOr returning"); run<K, 1>("ds_read + ds_or (no return)"); run<K, 2>("ds_read only");
    ROW(1) ROW(2) ROW(4) ROW(8) ROW(16) ROW(32) ROW(64)
    return 0;
}
