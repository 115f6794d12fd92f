// Microbenchmark: what the bank conflicts of triple-product stage 2's LDS gather cost, and whether
// choosing the columns a wave reads in one step can remove them.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lds_conflict.hip -o lds_conflict && ./lds_conflict
// Tile [c][16 + 2 pad] doubles (the stage-2 layout): lane reads 8 x ds_read_b128 at column c.  The 16-byte
// slot of read j is (9 c + j) mod 16, so a 16-lane conflict group (MI355X_MICROARCH.md, LDS table) is
// conflict-free iff its columns are distinct mod 16.
//   MODE 0  random columns
//   MODE 1  columns distinct mod 16 inside each ds_read_b128 lane group {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32
//   MODE 2  columns distinct mod 16 inside each run of 16 consecutive lanes (the wrong grouping, as a control)
//   MODE 3  random columns, a random 66 % of the lanes take part (sliced-ELL padding)
//   MODE 4  MODE 1 with the same 66 % of the lanes
//   FMA 1   v_fma_f64 instead of v_mul_f64 + v_add_f64 (the library computes unfused, as the reference's x86 build does)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int NW, int FMA = 0>
__global__ __launch_bounds__(NW * 64) void k(int iters, double *out)
{
    extern __shared__ double tl[];
    constexpr int CH = 1024, LD = 18;
    for (int x = threadIdx.x; x < CH * LD; x += NW * 64) tl[x] = 1.0 + (x & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int l32 = lane & 31;
    // position of the lane inside its ds_read_b128 group
    int pos;
    if (l32 < 4) pos = l32; else if (l32 < 12) pos = l32 - 4; else if (l32 < 16) pos = l32 - 8;
    else if (l32 < 20) pos = l32 - 8; else if (l32 < 28) pos = l32 - 12; else pos = l32 - 16;
    if (MODE == 2) pos = lane & 15;
    unsigned s = (blockIdx.x * NW * 64 + threadIdx.x) * 2654435761u + 12345u;
    double sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] = 0.0;
    const double hv = 1.0000001;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        int c = (s >> 10) & (CH - 1);
        if (MODE == 1 || MODE == 2 || MODE == 4) c = (c & ~15) | ((pos + it) & 15);
        if ((MODE == 3 || MODE == 4) && ((s >> 22) & 127) >= 84) continue;
        const double2 *p = (const double2 *)(tl + c * LD);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double2 v = p[r];
            if (FMA) { sum[2 * r] = __builtin_fma(v.x, hv, sum[2 * r]); sum[2 * r + 1] = __builtin_fma(v.y, hv, sum[2 * r + 1]); }
            else { sum[2 * r] += v.x * hv; sum[2 * r + 1] += v.y * hv; }
        }
    }
    double t = 0;
    for (int r = 0; r < 16; ++r) t += sum[r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = t;
}

template <int MODE, int NW, int FMA = 0>
void run(const char *name)
{
    const int grid = 256, iters = 20000;
    double *o; hipMalloc(&o, (size_t)grid * NW * 64 * 8);
    const size_t lds = 1024 * 18 * 8;
    hipFuncSetAttribute((const void *)k<MODE, NW, FMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NW, FMA><<<grid, NW * 64, lds>>>(100, o);
    hipEventRecord(e0);
    k<MODE, NW, FMA><<<grid, NW * 64, lds>>>(iters, o);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)grid * NW * 64 * iters * 16.0;
    printf("%-60s NW=%2d : %8.3f ms  %7.2f T gather-FMA/s (lane slots)  [%s]\n", name, NW, ms, fma / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
    hipFree(o);
}

int main()
{
    run<0, 16>("random columns");
    run<1, 16>("distinct mod 16 per b128 lane group");
    run<2, 16>("distinct mod 16 per 16 consecutive lanes (control)");
    run<3, 16>("random columns, 66 % of lanes");
    run<4, 16>("distinct per b128 group, 66 % of lanes");
    run<0, 16, 1>("random columns, fma");
    run<1, 16, 1>("distinct per b128 group, fma");
    run<4, 16, 1>("distinct per b128 group, 66 % of lanes, fma");
    run<1, 8, 1>("distinct per b128 group, fma");
    run<1, 8, 0>("distinct per b128 group");
    return 0;
}
