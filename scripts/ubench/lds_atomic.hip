// Microbenchmark: cost of one wave-wide LDS f64 accumulate, three forms.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const int *idx, int iters, int nactive, double *out, long long *cyc)
{
    extern __shared__ double acc[];
    const int lane = threadIdx.x & 63;
    const int W = 4000;
    for (int x = threadIdx.x; x < W * NW; x += NW * 64) acc[x] = 0.0;
    __syncthreads();
    double *my = acc + (threadIdx.x >> 6) * W;
    // per-lane column sequence (pre-generated, distinct inside a wave-instruction)
    int c[16];
    for (int u = 0; u < 16; ++u) c[u] = idx[(blockIdx.x * 16 + u) * 64 + lane];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (lane < nactive) {
                const double p = 1.0 + u;
                if (MODE == 0) (void)__hip_atomic_fetch_add(&my[c[u]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 1) my[c[u]] = my[c[u]] + p;
                if (MODE == 2) { volatile double *q = &my[c[u]]; *q = *q + p; }
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    double s = 0;
    for (int x = lane; x < W; x += 64) s += my[x];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int MODE, int NW>
void run(const char *name, int nactive, int wgs_per_cu)
{
    const int grid = 256 * wgs_per_cu, iters = 2000;
    std::vector<int> h(grid * 16 * 64);
    for (int b = 0; b < grid * 16; ++b) {           // distinct columns within each wave-instruction
        for (int l = 0; l < 64; ++l) {
            int v;
            bool ok;
            do { v = rand() % 4000; ok = true; for (int m = 0; m < l; ++m) if (h[b * 64 + m] == v) ok = false; } while (!ok);
            h[b * 64 + l] = v;
        }
    }
    int *d; double *o; long long *cy;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, grid * NW * 64 * 8); hipMalloc(&cy, grid * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t lds = 4000 * 8 * NW;
    hipFuncSetAttribute((const void *)k<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE, NW><<<grid, NW * 64, lds>>>(d, 10, nactive, o, cy);
    hipEventRecord(e0);
    k<MODE, NW><<<grid, NW * 64, lds>>>(d, iters, nactive, o, cy);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(grid); hipMemcpy(hc.data(), cy, grid * 8, hipMemcpyDeviceToHost);
    double per = (double)hc[0] / (iters * 16.0);
    double adds = (double)grid * NW * iters * 16.0 * nactive;
    printf("%-28s NW=%d wg/cu=%d active=%2d : %7.1f clk64 ticks per wave-instr (block 0), %.3f ms, %.2f G adds/s chip\n", name, NW,
           wgs_per_cu, nactive, per, ms, adds / ms / 1e6);
    hipFree(d); hipFree(o); hipFree(cy);
}

int main()
{
    for (int na : {64, 40, 16}) {
        run<0, 1>("ds_add_f64 (atomic)", na, 4);
        run<1, 1>("read-add-write (plain)", na, 4);
        run<2, 1>("read-add-write (volatile)", na, 4);
    }
    run<0, 4>("ds_add_f64 (atomic)", 40, 1);
    run<1, 4>("read-add-write (plain)", 40, 1);
    run<0, 8>("ds_add_f64 (atomic)", 40, 1);
    run<1, 8>("read-add-write (plain)", 40, 1);
    return 0;
}
