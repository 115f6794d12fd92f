// Microbenchmark: bank conflicts of ds_add_f64 (the accumulate of smm_numeric) -- which lanes collide, and what a
// conflict-free order of a piece's entries would buy.
//   hipcc --offload-arch=gfx950 -O3 lds_add_conflict.hip -o lds_add_conflict && ./lds_add_conflict
// 16 waves per workgroup add into one tile of 16384 doubles, 16 adds per lane and round, columns per mode:
//   0  random
//   1  lane (consecutive: one pass per 32 lanes at best)
//   2  random, but distinct mod 32 inside each 32-lane half
//   3  random, but distinct mod 32 inside each run of 16 lanes (two lanes per slot and half)
//   4  random, but distinct mod 64 over the wave
//   5  all lanes of a 32-lane half on ONE slot (different addresses): the worst case
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, double *out)
{
    extern __shared__ double acc[];
    constexpr int W = 16384;
    for (int x = threadIdx.x; x < W; x += 1024) acc[x] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            s = s * 1664525u + 1013904223u;
            int c = (s >> 9) & (W - 1);
            if (MODE == 1) c = (lane + 64 * u + it) & (W - 1);
            if (MODE == 2) c = (c & ~31) | ((lane + u) & 31);
            if (MODE == 3) c = (c & ~31) | (((lane & 15) * 2 + (u & 1)) & 31);
            if (MODE == 4) c = (c & ~63) | ((lane + u) & 63);
            if (MODE == 5) c = (c & ~31) | 7;
            (void)__hip_atomic_fetch_add(&acc[c], 1.0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    double t = 0;
    for (int x = threadIdx.x; x < W; x += 1024) t += acc[x];
    out[blockIdx.x * 1024 + threadIdx.x] = t;
}

template <int MODE>
void run(const char *name)
{
    const int grid = 256, iters = 2000;
    double *o; hipMalloc(&o, (size_t)grid * 1024 * 8);
    const size_t lds = 16384 * 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 1024, lds>>>(10, o);
    hipEventRecord(e0);
    k<MODE><<<grid, 1024, lds>>>(iters, o);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double adds = (double)grid * 1024 * iters * 16.0;
    printf("%-58s %8.3f ms  %7.2f T adds/s  %5.1f cycles per wave instruction and CU at 2.4 GHz  [%s]\n", name, ms, adds / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (16.0 * iters * 16.0), hipGetErrorString(hipGetLastError()));
    hipFree(o);
}

int main()
{
    run<0>("random columns");
    run<1>("consecutive columns");
    run<2>("distinct mod 32 per 32-lane half");
    run<3>("distinct mod 32 per 16 lanes");
    run<4>("distinct mod 64 over the wave");
    run<5>("one slot per half (worst case)");
    return 0;
}
