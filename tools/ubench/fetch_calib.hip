// What do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report on gfx950 for the access shapes the codec kernels use?
// Streams a 1 GiB buffer (four times the 256 MB Infinity Cache) once per kernel with 1 / 2 / 4 / 8 / 16 bytes per lane,
// with a "half line" pattern (each wave uses 64 bytes of every 128-byte line: what a tile whose pixel rows straddle lines
// does), and writes with 4 / 8 / 16 bytes per lane and in 64-byte half lines.  Run it plainly for GB/s per shape, and under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib        (then WRITE_SIZE, then the raw TCC request counters)
// for the bytes the counters report per kernel; tools/collect_roof.sh does both and keeps the table under profiles/.
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned __attribute__((ext_vector_type(2))) u2;
typedef unsigned __attribute__((ext_vector_type(4))) u4;

template <typename T> __device__ __forceinline__ unsigned fold(T v);
template <> __device__ __forceinline__ unsigned fold<uint8_t>(uint8_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold<uint16_t>(uint16_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold<unsigned>(unsigned v) { return v; }
template <> __device__ __forceinline__ unsigned fold<u2>(u2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ unsigned fold<u4>(u4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// every lane reads sizeof(T) bytes, a wave a contiguous run of 64 * sizeof(T) bytes, eight runs per thread
template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ p, size_t n, unsigned *out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc ^= fold<T>(__builtin_nontemporal_load(p + i));
    if (acc == 0x12345u) out[0] = acc;
}
// half lines: 16 lanes x 4 bytes = the first (HI = 0) or second (HI = 1) 64 bytes of each 128-byte line
template <int HI>
__global__ __launch_bounds__(256) void k_read_half(const unsigned *__restrict__ p, size_t nlines, unsigned *out)
{
    const size_t stride = (size_t)gridDim.x * 16;
    unsigned acc = 0;
    for (size_t l = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); l < nlines; l += stride) acc ^= p[l * 32 + HI * 16 + (threadIdx.x & 15)];
    if (acc == 0x12345u) out[0] = acc;
}
// 32-byte sectors: 8 lanes x 4 bytes of each 128-byte line
__global__ __launch_bounds__(256) void k_read_quarter(const unsigned *__restrict__ p, size_t nlines, unsigned *out)
{
    const size_t stride = (size_t)gridDim.x * 32;
    unsigned acc = 0;
    for (size_t l = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); l < nlines; l += stride) acc ^= p[l * 32 + (threadIdx.x & 7)];
    if (acc == 0x12345u) out[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ p, size_t n, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    T val; __builtin_memset(&val, (int)v, sizeof(T));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = val;
}
__global__ __launch_bounds__(256) void k_write_half(unsigned *__restrict__ p, size_t nlines, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 16;
    for (size_t l = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); l < nlines; l += stride) p[l * 32 + (threadIdx.x & 15)] = v;
}

static hipEvent_t e0, e1;
#define TIME(name, useful, ...) do { \
    __VA_ARGS__; (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); __VA_ARGS__; (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); \
    printf("%-34s %8.3f ms  %7.1f GB/s of the bytes used (%.1f MB)\n", name, ms, (useful) / (ms * 1e-3) / 1e9, (useful) / 1e6); } while (0)

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *d; unsigned *o;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 64);
    (void)hipMemset(d, 1, bytes);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int g = 256 * 32;
    TIME("k_read<16 B/lane>", (double)bytes, hipLaunchKernelGGL(k_read<u4>, dim3(g), dim3(256), 0, 0, (const u4 *)d, bytes / 16, o));
    TIME("k_read<8 B/lane>", (double)bytes, hipLaunchKernelGGL(k_read<u2>, dim3(g), dim3(256), 0, 0, (const u2 *)d, bytes / 8, o));
    TIME("k_read<4 B/lane>", (double)bytes, hipLaunchKernelGGL(k_read<unsigned>, dim3(g), dim3(256), 0, 0, (const unsigned *)d, bytes / 4, o));
    TIME("k_read<2 B/lane>", (double)bytes / 2, hipLaunchKernelGGL(k_read<uint16_t>, dim3(g), dim3(256), 0, 0, (const uint16_t *)d, bytes / 4, o));
    TIME("k_read<1 B/lane>", (double)bytes / 4, hipLaunchKernelGGL(k_read<uint8_t>, dim3(g), dim3(256), 0, 0, (const uint8_t *)d, bytes / 4, o));
    TIME("k_read_half<0> (64 B of each line)", (double)bytes / 2, hipLaunchKernelGGL(k_read_half<0>, dim3(g), dim3(256), 0, 0, (const unsigned *)d, bytes / 128, o));
    TIME("k_read_half<1> (the other 64 B)", (double)bytes / 2, hipLaunchKernelGGL(k_read_half<1>, dim3(g), dim3(256), 0, 0, (const unsigned *)d, bytes / 128, o));
    TIME("k_read_quarter (32 B of each line)", (double)bytes / 4, hipLaunchKernelGGL(k_read_quarter, dim3(g), dim3(256), 0, 0, (const unsigned *)d, bytes / 128, o));
    TIME("k_write<16 B/lane>", (double)bytes, hipLaunchKernelGGL(k_write<u4>, dim3(g), dim3(256), 0, 0, (u4 *)d, bytes / 16, 3u));
    TIME("k_write<8 B/lane>", (double)bytes, hipLaunchKernelGGL(k_write<u2>, dim3(g), dim3(256), 0, 0, (u2 *)d, bytes / 8, 3u));
    TIME("k_write<4 B/lane>", (double)bytes, hipLaunchKernelGGL(k_write<unsigned>, dim3(g), dim3(256), 0, 0, (unsigned *)d, bytes / 4, 3u));
    TIME("k_write<2 B/lane>", (double)bytes / 2, hipLaunchKernelGGL(k_write<uint16_t>, dim3(g), dim3(256), 0, 0, (uint16_t *)d, bytes / 4, 3u));
    TIME("k_write_half (64 B of each line)", (double)bytes / 2, hipLaunchKernelGGL(k_write_half, dim3(g), dim3(256), 0, 0, (unsigned *)d, bytes / 128, 3u));
    return 0;
}
