// Does ds_add_f64 apply the lanes of ONE wave-instruction that hit the same address in ascending
// lane order?  (The ISA does not promise it.)  Random values spanning many magnitudes, random
// groupings of lanes onto addresses; the result is compared bit for bit with a CPU sum in lane order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

__global__ void k(const int *addr, const double *val, const double *init, double *out, int ntrial, int naddr)
{
    extern __shared__ double acc[];
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < ntrial; t += gridDim.x) {
        for (int x = lane; x < naddr; x += 64) acc[x] = init[t * naddr + x];
        __builtin_amdgcn_wave_barrier();
        const int a = addr[t * 64 + lane];
        if (a >= 0) (void)__hip_atomic_fetch_add(&acc[a], val[t * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        for (int x = lane; x < naddr; x += 64) out[t * naddr + x] = acc[x];
        __builtin_amdgcn_wave_barrier();
    }
}

int main()
{
    const int ntrial = 200000, naddr = 32;
    std::vector<int> addr(ntrial * 64);
    std::vector<double> val(ntrial * 64), init(ntrial * naddr), want(ntrial * naddr), got(ntrial * naddr);
    srand(12345);
    for (int t = 0; t < ntrial; ++t) {
        const int spread = 1 + rand() % naddr;              // how many distinct addresses this trial uses
        for (int x = 0; x < naddr; ++x) init[t * naddr + x] = want[t * naddr + x] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 60 - 30);
        for (int l = 0; l < 64; ++l) {
            addr[t * 64 + l] = (rand() % 8 == 0) ? -1 : rand() % spread;
            val[t * 64 + l] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 60 - 30);
        }
        for (int l = 0; l < 64; ++l) if (addr[t * 64 + l] >= 0) want[t * naddr + addr[t * 64 + l]] += val[t * 64 + l];
    }
    int *da; double *dv, *di, *dout;
    hipMalloc(&da, addr.size() * 4); hipMalloc(&dv, val.size() * 8); hipMalloc(&di, init.size() * 8); hipMalloc(&dout, got.size() * 8);
    hipMemcpy(da, addr.data(), addr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, val.data(), val.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(di, init.data(), init.size() * 8, hipMemcpyHostToDevice);
    k<<<2048, 64, naddr * 8>>>(da, dv, di, dout, ntrial, naddr);
    hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0, total = 0;
    for (size_t i = 0; i < got.size(); ++i) { ++total; if (memcmp(&got[i], &want[i], 8)) ++bad; }
    // how often does ANY other order differ?  (descending lane order as a control)
    long ctrl = 0;
    for (int t = 0; t < ntrial; ++t) {
        std::vector<double> w(init.begin() + t * naddr, init.begin() + (t + 1) * naddr);
        for (int l = 63; l >= 0; --l) if (addr[t * 64 + l] >= 0) w[addr[t * 64 + l]] += val[t * 64 + l];
        for (int x = 0; x < naddr; ++x) if (memcmp(&w[x], &want[t * naddr + x], 8)) ++ctrl;
    }
    printf("accumulators compared: %ld, differing from ascending-lane-order sum: %ld  (control: descending order differs in %ld)\n", total, bad, ctrl);
    return bad != 0;
}
