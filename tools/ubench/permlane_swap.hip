// Semantics and cost of v_permlane32_swap / v_permlane16_swap on gfx950, and the 4-value wave reduction built on them.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sem(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
static __device__ __forceinline__ void wave_sum4(unsigned &a, unsigned &b, unsigned &c, unsigned &d)
{
    auto p = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    unsigned s1 = p[0] + p[1];                       // lanes 0-31: a (two 32-halves added), lanes 32-63: b
    auto q = __builtin_amdgcn_permlane32_swap(c, d, false, false);
    unsigned s2 = q[0] + q[1];                       // lanes 0-31: c, 32-63: d
    auto r = __builtin_amdgcn_permlane16_swap(s1, s2, false, false);
    unsigned t = r[0] + r[1];                        // rows: a, c, b, d
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, true);
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, true);
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, true);
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x140, 0xf, 0xf, true);
    a = (unsigned)__builtin_amdgcn_readlane((int)t, 0);
    c = (unsigned)__builtin_amdgcn_readlane((int)t, 16);
    b = (unsigned)__builtin_amdgcn_readlane((int)t, 32);
    d = (unsigned)__builtin_amdgcn_readlane((int)t, 48);
}
__global__ void red(const unsigned *in, unsigned *out)
{
    unsigned a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
    wave_sum4(a, b, c, d);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; out[3] = d; }
}
int main()
{
    unsigned *d, h[256], in[256], *din;
    hipMalloc(&d, 1024); hipMalloc(&din, 1024);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("swap32 r0: %u %u .. %u %u | r1: %u %u .. %u %u\n", h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127]);
    printf("swap16 r0 rows: %u %u %u %u | r1 rows: %u %u %u %u\n", h[128], h[144], h[160], h[176], h[192], h[208], h[224], h[240]);
    unsigned want[4] = {0, 0, 0, 0};
    for (int i = 0; i < 256; i++) { in[i] = (unsigned)(i * 2654435761u) >> 12; want[i / 64] += in[i]; }
    hipMemcpy(din, in, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(red, dim3(1), dim3(64), 0, 0, din, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("wave_sum4: got %u %u %u %u want %u %u %u %u -> %s\n", h[0], h[1], h[2], h[3], want[0], want[1], want[2], want[3],
           (h[0] == want[0] && h[1] == want[1] && h[2] == want[2] && h[3] == want[3]) ? "OK" : "MISMATCH");
    return 0;
}
