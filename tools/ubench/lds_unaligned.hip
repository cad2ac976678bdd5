// Does ds_read_b32 at a byte address that is not a multiple of 4 return the four bytes at that address on gfx950 (unaligned access
// mode), and what does it cost?  The motion search re-aligns every reference dword it reads from its staged window with
// v_alignbyte_b32 (two aligned reads + one VALU per dword); an unaligned read would be one LDS instruction.  Output kept under profiles/.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_check(unsigned *out, int shift)
{
    __shared__ __attribute__((aligned(16))) unsigned char b[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) b[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(b) + 4u * threadIdx.x + (unsigned)shift;      // LDS byte address (low 32 bits of the shared pointer)
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v;
}
template <int SHIFT, bool ALIGNBYTE>
__global__ void k_time(unsigned *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char b[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) b[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned acc = 0;
    unsigned addr = (unsigned)(size_t)(b) + 4u * (threadIdx.x & 63) + 96u * (threadIdx.x >> 6);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (ALIGNBYTE) {
                unsigned lo, hi;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(*(unsigned long long *)&lo) : "v"(addr + 96u * u));
                (void)hi;
                unsigned long long w; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr + 96u * u) : "memory");
                acc += __builtin_amdgcn_alignbyte((unsigned)(w >> 32), (unsigned)w, (unsigned)SHIFT);
            } else {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + 96u * u + (unsigned)SHIFT) : "memory");
                acc += v;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 1 << 22);
    std::vector<unsigned> h(64);
    int bad = 0;
    for (int sh = 0; sh < 4; sh++) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, sh);
        (void)hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            unsigned want = 0;
            for (int k = 0; k < 4; k++) want |= (unsigned)(unsigned char)((4 * l + sh + k) * 7 + 3) << (8 * k);
            if (h[l] != want) { if (bad < 4) printf("shift %d lane %d: got %08x want %08x\n", sh, l, h[l], want); bad++; }
        }
    }
    printf("ds_read_b32 at byte offsets 0..3: %s (%d mismatches)\n", bad ? "WRONG" : "returns the bytes at the address", bad);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name) {
        hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 256);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 2048);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %.3f ms for %.1f G lane-dwords -> %.1f G dwords/s\n", name, ms, 256.0 * 8 * 256 * 2048 * 16 / 1e9, 256.0 * 8 * 256 * 2048 * 16 / ms / 1e6);
    };
    run(k_time<0, false>, "ds_read_b32 aligned");
    run(k_time<1, false>, "ds_read_b32 at byte offset 1");
    run(k_time<2, false>, "ds_read_b32 at byte offset 2");
    run(k_time<3, false>, "ds_read_b32 at byte offset 3");
    run(k_time<1, true>, "ds_read2_b32 + v_alignbyte_b32 (today's form)");
    return 0;
}
