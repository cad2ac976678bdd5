// Issue rate of the integer / byte VALU instructions (and the scalar unit) the codec kernels lean on, on gfx950, as a
// function of the number of waves resident on a SIMD.  Settles what a wave64 VALU instruction costs: the chip's SIMDs are
// 32 lanes wide, so one wave alone issues a VALU instruction every 4 cycles and two or more waves on the SIMD share a
// 2-cycle slot -- IF that holds for the integer forms used here.  Output is kept under profiles/.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
//
// Occupancy is pinned by construction: one workgroup of 256*W threads per CU (W = waves per SIMD; W = 8: two workgroups of
// 1024), each asking for enough LDS that no further workgroup fits beside it; every wave records the SIMD it ran on
// (HW_ID) and its own cycle count (s_memtime), so the table shows the waves per SIMD that were really there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define ITERS 4096
#define UNROLL 8

enum { OP_ADD, OP_SAD_U8, OP_SAD_U16, OP_ALIGNBYTE, OP_PERM, OP_DOT4, OP_MUL_LO, OP_MUL_U24, OP_MAD_U24, OP_PK_ADD_I16, OP_PK_MUL_LO_U16,
       OP_PK_MAX_I16, OP_PK_ASHR_I16, OP_MAX_I32, OP_MED3_I32, OP_BFE, OP_LSHL_ADD, OP_ADD3, OP_CNDMASK, OP_DPP_ADD, OP_READLANE,
       OP_SALU, OP_VALU_SALU, OP_VALU_SALU2, OP_LSHL_ADD_U64, OP_MOV, OP_SAT_PK_U8, OP_CNDMASK_VCCSET, OP_CNDMASK_SGPR, OP_CNDMASK_NODEP,
       OP_SAD_SALU, OP_CND_E64_VCC, OP_CND_MIX4, OP_ADDC, OP_CMP_VCC, OP_CMP_SGPR, OP_CND_VCC_S_MOV, OP_CND_MIX_CMP, OP_SUB, OP_AND, OP_LSHLREV, OP_ASHRREV, OP_BFI, OP_AND_OR, OP_MIN_U32, OP_PK_SUB_I16, OP_PK_LSHL, OP_PK_MAD_I16, OP_MAD_I32_I24, OP_XOR, OP_QSAD, OP_MQSAD, OP_MSAD_U8, OP_SAD_HI_U8, OP_OR, OP_LSHRREV, OP_PK_MIN_I16, OP_MED3_I16, OP_ADD_U16, OP_LERP_U8, OP_PK_ADD_U16_CLAMP, OP_QSAD_INDEP, OP_N };
static const char *names[OP_N] = { "v_add_u32", "v_sad_u8", "v_sad_u16", "v_alignbyte_b32", "v_perm_b32", "v_dot4_u32_u8", "v_mul_lo_u32",
       "v_mul_u32_u24", "v_mad_u32_u24", "v_pk_add_i16", "v_pk_mul_lo_u16", "v_pk_max_i16", "v_pk_ashrrev_i16", "v_max_i32", "v_med3_i32",
       "v_bfe_u32", "v_lshl_add_u32", "v_add3_u32", "v_cndmask_b32", "v_add_u32 dpp row_shr:1", "v_readlane_b32 (+s use)",
       "s_add_u32 alone", "v_add_u32 + s_add_u32 1:1", "v_add_u32 + 2 s_add_u32", "v_lshl_add_u64", "v_mov_b32", "v_sat_pk_u8_i16", "v_cndmask_b32 (vcc from v_cmp)", "v_cndmask_b32_e64 (sgpr pair)",
       "v_cndmask_b32 (no dependence)", "v_sad_u8 + s_add_u32 1:1", "v_cndmask_b32_e64 (vcc)", "1 v_cndmask e32 vcc + 3 v_add_u32", "v_addc_co_u32 (vcc in/out)",
       "v_cmp_gt_u32 -> vcc", "v_cmp_gt_u32_e64 -> sgpr pair", "v_cndmask_b32 e32 (vcc = s_mov -1)", "v_cmp -> vcc + v_cndmask e32 vcc", "v_sub_u32", "v_and_b32", "v_lshlrev_b32",
       "v_ashrrev_i32", "v_bfi_b32", "v_and_or_b32", "v_min_u32", "v_pk_sub_i16", "v_pk_lshlrev_b16", "v_pk_mad_i16", "v_mad_i32_i24", "v_xor_b32",
       "v_qsad_pk_u16_u8 (acc chain)", "v_mqsad_pk_u16_u8 (acc chain)", "v_msad_u8", "v_sad_hi_u8", "v_or_b32", "v_lshrrev_b32", "v_pk_min_i16", "v_med3_i16", "v_add_u16",
       "v_lerp_u8", "v_pk_add_u16 clamp", "v_qsad_pk_u16_u8 (acc = 0)" };

#define A1(s) asm volatile(s : "+v"(a[i]) : "v"(b), "v"(c))
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned long long *cyc, unsigned *hwid, unsigned seed, int iters)
{
    extern __shared__ unsigned lds[];
    unsigned a[UNROLL];
    unsigned long long w[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed * (threadIdx.x + i + 1); w[i] = a[i]; }
    unsigned b = seed ^ threadIdx.x, c = seed + 7;
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    if (seed == 0x7fffffffu) lds[threadIdx.x] = b;              // keeps the allocation
    __syncthreads();
    unsigned long long m64 = 0x5555555555555555ull * seed;
    const unsigned long long m64v = 0x0102030405060708ull * (seed + threadIdx.x);
    if (OP == OP_CNDMASK_VCCSET) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
    if (OP == OP_CND_VCC_S_MOV) asm volatile("s_mov_b64 vcc, -1" : : : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == OP_ADD) A1("v_add_u32 %0, %0, %1");
            if (OP == OP_SAD_U8) A1("v_sad_u8 %0, %0, %1, %2");
            if (OP == OP_SAD_U16) A1("v_sad_u16 %0, %0, %1, %2");
            if (OP == OP_ALIGNBYTE) A1("v_alignbyte_b32 %0, %0, %1, %2");
            if (OP == OP_PERM) A1("v_perm_b32 %0, %0, %1, %2");
            if (OP == OP_DOT4) A1("v_dot4_u32_u8 %0, %0, %1, %2");
            if (OP == OP_MUL_LO) A1("v_mul_lo_u32 %0, %0, %1");
            if (OP == OP_MUL_U24) A1("v_mul_u32_u24 %0, %0, %1");
            if (OP == OP_MAD_U24) A1("v_mad_u32_u24 %0, %0, %1, %2");
            if (OP == OP_PK_ADD_I16) A1("v_pk_add_i16 %0, %0, %1");
            if (OP == OP_PK_MUL_LO_U16) A1("v_pk_mul_lo_u16 %0, %0, %1");
            if (OP == OP_PK_MAX_I16) A1("v_pk_max_i16 %0, %0, %1");
            if (OP == OP_PK_ASHR_I16) A1("v_pk_ashrrev_i16 %0, 1, %0");
            if (OP == OP_MAX_I32) A1("v_max_i32 %0, %0, %1");
            if (OP == OP_MED3_I32) A1("v_med3_i32 %0, %0, %1, %2");
            if (OP == OP_BFE) A1("v_bfe_u32 %0, %0, 8, 8");
            if (OP == OP_LSHL_ADD) A1("v_lshl_add_u32 %0, %0, 1, %1");
            if (OP == OP_ADD3) A1("v_add3_u32 %0, %0, %1, %2");
            if (OP == OP_CNDMASK) A1("v_cndmask_b32 %0, %0, %1, vcc");
            if (OP == OP_DPP_ADD) A1("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf");
            if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            if (OP == OP_SAT_PK_U8) A1("v_sat_pk_u8_i16 %0, %0");
            if (OP == OP_READLANE) { asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a[i])); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s0)); }
            if (OP == OP_SALU) { asm volatile("s_add_u32 %0, %0, %1" : "+s"((i & 1) ? s0 : s2) : "s"(s1)); }
            if (OP == OP_VALU_SALU) { A1("v_add_u32 %0, %0, %1"); asm volatile("s_add_u32 %0, %0, %1" : "+s"((i & 1) ? s0 : s2) : "s"(s1)); }
            if (OP == OP_VALU_SALU2) { A1("v_add_u32 %0, %0, %1"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3)); }
            if (OP == OP_CNDMASK_VCCSET) A1("v_cndmask_b32 %0, %0, %1, vcc");
            if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m64));
            if (OP == OP_CNDMASK_NODEP) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
            if (OP == OP_SAD_SALU) { A1("v_sad_u8 %0, %0, %1, %2"); asm volatile("s_add_u32 %0, %0, %1" : "+s"((i & 1) ? s0 : s2) : "s"(s1)); }
            if (OP == OP_CND_E64_VCC) A1("v_cndmask_b32_e64 %0, %0, %1, vcc");
            if (OP == OP_CND_MIX4) { if ((i & 3) == 0) A1("v_cndmask_b32 %0, %0, %1, vcc"); else A1("v_add_u32 %0, %0, %1"); }
            if (OP == OP_ADDC) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CMP_VCC) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == OP_CMP_SGPR) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m64) : "v"(a[i]), "v"(b));
            if (OP == OP_CND_VCC_S_MOV) A1("v_cndmask_b32 %0, %0, %1, vcc");
            if (OP == OP_CND_MIX_CMP) { if (i & 1) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); else A1("v_cndmask_b32 %0, %0, %1, vcc"); }
            if (OP == OP_SUB) A1("v_sub_u32 %0, %0, %1");
            if (OP == OP_AND) A1("v_and_b32 %0, %0, %1");
            if (OP == OP_XOR) A1("v_xor_b32 %0, %0, %1");
            if (OP == OP_LSHLREV) A1("v_lshlrev_b32 %0, 1, %0");
            if (OP == OP_ASHRREV) A1("v_ashrrev_i32 %0, 1, %0");
            if (OP == OP_BFI) A1("v_bfi_b32 %0, %0, %1, %2");
            if (OP == OP_AND_OR) A1("v_and_or_b32 %0, %0, %1, %2");
            if (OP == OP_MIN_U32) A1("v_min_u32 %0, %0, %1");
            if (OP == OP_PK_SUB_I16) A1("v_pk_sub_i16 %0, %0, %1");
            if (OP == OP_PK_LSHL) A1("v_pk_lshlrev_b16 %0, 1, %0");
            if (OP == OP_PK_MAD_I16) A1("v_pk_mad_i16 %0, %0, %1, %2");
            if (OP == OP_MAD_I32_I24) A1("v_mad_i32_i24 %0, %0, %1, %2");
            if (OP == OP_QSAD) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(m64v), "v"(b));
            if (OP == OP_MQSAD) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(m64v), "v"(b));
            if (OP == OP_QSAD_INDEP) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, 0" : "=&v"(w[i]) : "v"(m64v), "v"(b));
            if (OP == OP_MSAD_U8) A1("v_msad_u8 %0, %0, %1, %2");
            if (OP == OP_SAD_HI_U8) A1("v_sad_hi_u8 %0, %0, %1, %2");
            if (OP == OP_OR) A1("v_or_b32 %0, %0, %1");
            if (OP == OP_LSHRREV) A1("v_lshrrev_b32 %0, 1, %0");
            if (OP == OP_PK_MIN_I16) A1("v_pk_min_i16 %0, %0, %1");
            if (OP == OP_MED3_I16) A1("v_med3_i16 %0, %0, %1, %2");
            if (OP == OP_ADD_U16) A1("v_add_u16 %0, %0, %1");
            if (OP == OP_LERP_U8) A1("v_lerp_u8 %0, %0, %1, %2");
            if (OP == OP_PK_ADD_U16_CLAMP) A1("v_pk_add_u16 %0, %0, %1 clamp");
            if (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(w[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned r = s0 ^ s1 ^ s2 ^ s3;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ (unsigned)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        const unsigned wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        unsigned id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cyc[wv] = t1 - t0;
        hwid[wv] = (id & 0xfff0u) | ((xcc & 0xf) << 16);    // simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13, xcc 19:16
    }
}

struct Res { double ms, ginstr, cyc_per_instr_wave, waves_per_simd; };
static int per_of(int op) { return (op == OP_VALU_SALU || op == OP_SAD_SALU || op == OP_READLANE) ? 2 : (op == OP_VALU_SALU2) ? 3 : 1; }
static int iters_of(int op) { return (op == OP_SALU || op == OP_READLANE) ? ITERS * 16 : ITERS; }
template <int OP> Res run(int W, unsigned *d, unsigned long long *dc, unsigned *dh, int ncu)
{
    const int bt = W >= 4 ? 1024 : 256 * W, per_cu = W >= 8 ? W / 4 : 1;
    const int blocks = ncu * per_cu, lds = per_cu == 1 ? 96 * 1024 : (160 * 1024 / per_cu) - 8 * 1024 > 64 * 1024 ? 72 * 1024 : 64 * 1024;
    (void)hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(bt), lds, 0, d, dc, dh, 3u, iters_of(OP));
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(bt), lds, 0, d, dc, dh, 3u, iters_of(OP));
    (void)hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed (W=%d lds=%d): %s\n", W, lds, hipGetErrorString(hipGetLastError())); exit(1); }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * bt / 64;
    std::vector<unsigned long long> hc(nw); std::vector<unsigned> hh(nw);
    (void)hipMemcpy(hc.data(), dc, nw * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hh.data(), dh, nw * 4, hipMemcpyDeviceToHost);
    std::sort(hh.begin(), hh.end());
    const int simds = (int)(std::unique(hh.begin(), hh.end()) - hh.begin());
    std::sort(hc.begin(), hc.end());
    const double n = (double)iters_of(OP) * UNROLL * per_of(OP);
    Res r; r.ms = ms; r.ginstr = nw * n / (ms * 1e-3) / 1e9; r.cyc_per_instr_wave = (double)hc[nw / 2] / n; r.waves_per_simd = (double)nw / simds;
    return r;
}

template <int OP> void sweep(unsigned *d, unsigned long long *dc, unsigned *dh, int ncu, double mhz)
{
    printf("%-28s", names[OP]);
    const int Ws[5] = { 1, 2, 3, 4, 8 };
    for (int w = 0; w < 5; w++) {
        Res r = run<OP>(Ws[w], d, dc, dh, ncu);
        // chip-wide G wave-instr/s | median s_memtime ticks per instruction of ONE wave | cycles of SIMD time per instruction = ms * clock / (instr per SIMD)
        const double simd_cyc = (r.ms * 1e-3) * (mhz * 1e6) / ((double)iters_of(OP) * UNROLL * per_of(OP) * r.waves_per_simd);
        printf(" | W=%d(%.2f) %7.1f G/s %5.2f cyc/SIMD %5.3f tick/wave-instr", Ws[w], r.waves_per_simd, r.ginstr, simd_cyc, r.cyc_per_instr_wave);
    }
    printf("\n");
}
template <int OP> struct All { static void go(unsigned *d, unsigned long long *dc, unsigned *dh, int ncu, double mhz) { sweep<OP>(d, dc, dh, ncu, mhz); All<OP + 1>::go(d, dc, dh, ncu, mhz); } };
template <> struct All<OP_N> { static void go(unsigned *, unsigned long long *, unsigned *, int, double) {} };

int main(int argc, char **argv)
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
    printf("# %s, %d CUs, clockRate %.0f MHz; per column: waves per SIMD asked (measured from HW_ID), chip-wide G wave-instructions/s,\n"
           "# cycles of SIMD time per wave-instruction at clockRate (4 = one wave64 instruction per 4 cycles, 2 = per 2 cycles),\n# median s_memtime ticks of one wave per instruction of its own stream\n", p.name, ncu, mhz);
    printf("# peak if 4 cycles: %.1f G wave-instr/s ; if 2 cycles: %.1f\n", ncu * 4 * mhz / 4e3, ncu * 4 * mhz / 2e3);
    unsigned *d; unsigned long long *dc; unsigned *dh;
    (void)hipMalloc(&d, (size_t)ncu * 2 * 1024 * 4); (void)hipMalloc(&dc, (size_t)ncu * 32 * 8); (void)hipMalloc(&dh, (size_t)ncu * 32 * 4);
    if (argc > 1 && !strcmp(argv[1], "new")) All<OP_QSAD>::go(d, dc, dh, ncu, mhz);      // the forms added in round 4 only
    else All<0>::go(d, dc, dh, ncu, mhz);
    return 0;
}
