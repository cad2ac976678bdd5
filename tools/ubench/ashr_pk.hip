// v_ashr_pk_u8_i32 (gfx950): what does it leave in bits 31:16 of the destination, and which source goes to which byte?
// Measured on MI355X (ROCm 7.2): byte 0 = sat_u8(src0 >> n), byte 1 = sat_u8(src1 >> n), bits 31:16 KEEP what the
// destination register held ("deadff00" below) -- hipcc's instruction selection merges the result into wider values as
// if they were zero (k_fwd_mc_fast produced wrong residuals that way), so csrc/Makefile refuses objects that contain it.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/ashr_pk.hip -o tools/ubench/ashr_pk && ./tools/ubench/ashr_pk
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int *a, const int *b, unsigned *out)
{
    const int i = threadIdx.x;
    unsigned d = 0xdeadbeefu;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 8" : "+v"(d) : "v"(a[i]), "v"(b[i]));
    out[i] = d;
}
int main()
{
    const int n = 64;
    int ha[n], hb[n];
    unsigned ho[n];
    for (int i = 0; i < n; i++) { ha[i] = (i - 8) * 2000; hb[i] = (40 - i) * 3000; }
    int *a, *b; unsigned *o;
    (void)hipMalloc(&a, sizeof ha); (void)hipMalloc(&b, sizeof hb); (void)hipMalloc(&o, sizeof ho);
    (void)hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, a, b, o);
    (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad_hi = 0, bad_lo = 0;
    for (int i = 0; i < n; i++) {
        auto sat = [](int v) { v >>= 8; return v < 0 ? 0 : (v > 255 ? 255 : v); };
        const unsigned want = (unsigned)sat(ha[i]) | ((unsigned)sat(hb[i]) << 8);
        if ((ho[i] & 0xffffu) != want) bad_lo++;
        if ((ho[i] >> 16) != 0) bad_hi++;
        if (i < 6 || i > 58) printf("a %d b %d -> %08x want low %04x\n", ha[i], hb[i], ho[i], want);
    }
    printf("low halves wrong: %d, high halves non-zero: %d\n", bad_lo, bad_hi);
    return 0;
}
