// Where do the workgroups of a CU-masked stream run?  One launch per mask spec on a stream made by hipExtStreamCreateWithCUMask; every
// wave stores its (XCC_ID, HW_ID); the host prints the CUs seen per XCD / shader engine.  Evidence for the bit -> CU mapping that
// dsvg_pipe.hip's cu_mask_stream() assumes (bit i = a CU of XCD i mod 8).   build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void k_where(unsigned *out, long long ticks)
{
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);     // keep the slot so that the launch spreads over every allowed CU
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = id; out[2 * blockIdx.x + 1] = xcc; }
}
static void run(const char *name, const uint32_t *m)
{
    hipStream_t st;
    if (m) { if (hipExtStreamCreateWithCUMask(&st, 8, m) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; } }
    else (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int NB = 8192;
    unsigned *d; (void)hipMalloc(&d, NB * 8);
    hipLaunchKernelGGL(k_where, dim3(NB), dim3(256), 0, st, d, 2000LL);      // 20 us per workgroup
    (void)hipStreamSynchronize(st);
    std::vector<unsigned> h(2 * NB);
    (void)hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;          // xcc -> set of (se, sh, cu)
    for (int i = 0; i < NB; i++) {
        const unsigned id = h[2 * i], xcc = h[2 * i + 1] & 15;
        const int cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        per_xcc[(int)xcc].insert(se * 100 + sh * 16 + cu);
    }
    int tot = 0;
    printf("%s:", name);
    for (auto &kv : per_xcc) { printf("  xcc%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d CUs\n", tot);
    if (m && tot <= 64)
        for (auto &kv : per_xcc) { printf("    xcc%d (se*100+sh*16+cu):", kv.first); for (int v : kv.second) printf(" %d", v); printf("\n"); }
    (void)hipFree(d); (void)hipStreamDestroy(st);
}
int main()
{
    run("no mask", nullptr);
    uint32_t m[8];
    memset(m, 0, sizeof m); m[0] = 0xff; run("bits 0-7", m);
    memset(m, 0, sizeof m); m[0] = 0xffffffff; run("bits 0-31", m);
    memset(m, 0, sizeof m); m[0] = m[1] = 0xffffffff; run("bits 0-63", m);
    memset(m, 0, sizeof m); for (int i = 0; i < 8; i++) m[i] = 0x01010101; run("bits = 0 mod 8 (one XCD?)", m);
    memset(m, 0, sizeof m); for (int i = 0; i < 8; i++) m[i] = 0x03030303; run("bits = 0,1 mod 8 (two XCDs?)", m);
    memset(m, 0xff, sizeof m); m[0] = m[1] = 0; run("bits 64-255", m);
    return 0;
}
