// Checks the EXEC-narrowed load used by smm_symbolic (load_lanes_below): lanes >= cnt must keep their value.
//   hipcc --offload-arch=gfx950 -O3 exec_mask.hip -o exec_mask && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void load_lanes_below(int &v, const int *idx, int cnt, int lane4)
{
    const int sh = cnt >= 64 ? 0 : 64 - cnt;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_lshr_b64 exec, -1, %3\n\t"
                 "global_load_dword %0, %2, %4\n\t"
                 "s_mov_b64 exec, %1"
                 : "+v"(v), "=&s"(saved)
                 : "v"(lane4), "s"(sh), "s"(idx)
                 : "scc");
}

__global__ void k(const int *buf, const int *cnts, int ncnt, int *out)
{
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < ncnt; ++i) {
        const int cnt = __builtin_amdgcn_readfirstlane(cnts[i]);
        int v = -7;
        load_lanes_below(v, buf + i * 64, cnt, lane * 4);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
        out[i * 64 + lane] = v;
    }
}

int main()
{
    const int cn[] = {1, 4, 31, 32, 33, 63, 64, 100};
    const int n = sizeof(cn) / sizeof(int);
    int *buf, *cnts, *out, h[64 * 8], ho[64 * 8];
    for (int i = 0; i < 64 * n; ++i) h[i] = i;
    hipMalloc(&buf, sizeof(h)); hipMalloc(&cnts, sizeof(cn)); hipMalloc(&out, sizeof(ho));
    hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(cnts, cn, sizeof(cn), hipMemcpyHostToDevice);
    k<<<1, 64>>>(buf, cnts, n, out);
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        int loaded = 0, wrong = 0;
        for (int l = 0; l < 64; ++l) {
            const bool want = l < cn[i];
            if (ho[i * 64 + l] == i * 64 + l) ++loaded;
            if (ho[i * 64 + l] != (want ? i * 64 + l : -7)) ++wrong;
        }
        printf("cnt %3d: lanes loaded %2d, wrong %d\n", cn[i], loaded, wrong);
        bad += wrong;
    }
    printf(bad ? "BAD\n" : "OK\n");
    return bad != 0;
}
