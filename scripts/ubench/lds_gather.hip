// Microbenchmark: LDS gather shapes for triple-product stage 2 (8 B of LDS per multiply-add either way).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lds_gather.hip -o lds_gather && ./lds_gather
//   MODE 0  lanes own k, tile [r][chunk]:     8 x ds_read_b64 at random per-lane columns
//   MODE 1  lanes own k, tile [c][8]:         4 x ds_read_b128 (64 contiguous bytes per lane)
//   MODE 2  lanes own k, tile [c][8 + 2 pad]: same, 80-byte stride
//   MODE 3  lanes own i, tile [c][64 lanes]:  8 x ds_read_b64, column uniform in the wave (conflict-free)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

//   MASK 1: a random 57 % of the lanes take part in each step (sliced-ELL padding); MASK 2: lanes 0-35 only
template <int MODE, int NW, int MASK = 0>
__global__ __launch_bounds__(NW * 64) void k(int iters, double *out)
{
    extern __shared__ double tl[];
    constexpr int CH = 2048;                       // columns (MODE 0-2); MODE 3 uses 16384/64 = 256 columns
    constexpr int TOT = (MODE == 2) ? CH * 10 : CH * 8;
    for (int x = threadIdx.x; x < TOT; x += NW * 64) tl[x] = 1.0 + (x & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * NW * 64 + threadIdx.x) * 2654435761u + 12345u;
    unsigned su = __builtin_amdgcn_readfirstlane(s);
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double hv = 1.0000001;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            s = s * 1664525u + 1013904223u;
            const int c = (s >> 10) & (CH - 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) sum[r] += tl[r * CH + c] * hv;
        } else if (MODE == 1 || MODE == 2) {
            s = s * 1664525u + 1013904223u;
            const int c = (s >> 10) & (CH - 1);
            const double2 *p = (const double2 *)(tl + c * (MODE == 2 ? 10 : 8));
            if (MASK == 1 && ((s >> 22) & 127) >= 73) continue;
            if (MASK == 2 && lane >= 36) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = p[r];
                sum[2 * r] += v.x * hv;
                sum[2 * r + 1] += v.y * hv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                su = su * 1664525u + 1013904223u;
                const int c = (su >> 10) & 255;
                sum[r] += tl[c * 64 + lane] * hv;
            }
        }
    }
    double t = 0;
    for (int r = 0; r < 8; ++r) t += sum[r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = t;
}

template <int MODE, int NW, int MASK = 0>
void run(const char *name)
{
    const int grid = 256, iters = 20000;
    double *o; hipMalloc(&o, (size_t)grid * NW * 64 * 8);
    const size_t lds = (MODE == 2 ? 2048 * 10 : 2048 * 8) * 8;
    hipFuncSetAttribute((const void *)k<MODE, NW, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NW, MASK><<<grid, NW * 64, lds>>>(100, o);
    hipEventRecord(e0);
    k<MODE, NW, MASK><<<grid, NW * 64, lds>>>(iters, o);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)grid * NW * 64 * iters * 8.0;
    printf("%-44s NW=%2d : %8.3f ms  %7.2f T gather-FMA/s  (%5.1f TB/s of LDS)  [%s]\n", name, NW, ms, fma / ms / 1e9,
           fma * 8 / ms / 1e9, hipGetErrorString(hipGetLastError()));
    hipFree(o);
}

int main()
{
    run<0, 16>("lanes=k [r][chunk] 8 x b64 random");
    run<1, 16>("lanes=k [c][8] 4 x b128 random");
    run<2, 16>("lanes=k [c][10] 4 x b128 random, padded");
    run<3, 16>("lanes=i [c][64] 8 x b64 uniform column");
    run<2, 16, 1>("padded b128, random 57 % of lanes per step");
    run<2, 16, 2>("padded b128, lanes 0-35 only");
    run<0, 8>("lanes=k [r][chunk] 8 x b64 random");
    run<1, 8>("lanes=k [c][8] 4 x b128 random");
    run<3, 8>("lanes=i [c][64] 8 x b64 uniform column");
    return 0;
}
