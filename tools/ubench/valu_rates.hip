// Throughput of the byte/SAD VALU instructions the codec kernels lean on, relative to v_add_u32, on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    unsigned b = seed ^ threadIdx.x, c = seed + 7;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + b;
            if (OP == 1) a[i] = __builtin_amdgcn_sad_u8(a[i], b, c);
            if (OP == 2) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c);
            if (OP == 3) a[i] = (a[i] > b) ? c : a[i];
            if (OP == 4) a[i] = a[i] * b;
            if (OP == 5) a[i] = __builtin_amdgcn_udot4(a[i], b, c, false);
            if (OP == 6) a[i] += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[i], 0xB1, 0xf, 0xf, true);
            if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], b, c);
            if (OP == 8) a[i] = (unsigned)__builtin_amdgcn_readlane((int)a[i], 5) + b;
            if (OP == 9) a[i] = __builtin_amdgcn_ubfe(a[i], 8u, 8u) + c;
            if (OP == 10) a[i] = (unsigned)max((int)a[i], (int)b);
            if (OP == 11) a[i] = __builtin_amdgcn_sad_u16(a[i], b, c);
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> double run(unsigned *d, const char *name, double ref)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITERS * 8;          // wave-instructions
    const double rate = winstr / (ms * 1e-3) / 1024.0;             // per SIMD per second
    printf("%-22s %8.3f ms  %7.1f M wave-instr/s/SIMD  rel %.2f\n", name, ms, rate / 1e6, ref > 0 ? rate / ref : 1.0);
    return rate;
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const double r = run<0>(d, "v_add_u32", 0);
    run<1>(d, "v_sad_u8", r); run<11>(d, "v_sad_u16", r); run<2>(d, "v_alignbyte_b32", r); run<3>(d, "v_cmp+v_cndmask", r);
    run<4>(d, "v_mul_lo_u32", r); run<5>(d, "v_dot4_u32_u8", r); run<6>(d, "v_add_u32 dpp quad_perm", r);
    run<7>(d, "v_perm_b32", r); run<8>(d, "v_readlane+add", r); run<9>(d, "v_bfe_u32+add", r); run<10>(d, "v_max_i32", r);
    return 0;
}
