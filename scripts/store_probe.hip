// What separates a row-segmented store stream (67-70 % of the HBM peak) from a linear one (85 %)?
// Write-only kernels over a 16448 x 16387 fp64 array (the 16384^2 field at DL_ESM_ALIGNMENT=64), one 16-byte element per
// thread, 256-thread workgroups, non-temporal stores:
//   V0 linear       : thread t writes element t of the whole array (what fill_linear_k does)
//   V1 linear+mask  : the same indexing, but only columns 1..16446 of each row are stored (2 of 8224 pairs per row masked/partial)
//   V2 rowseg       : workgroup -> (row, segment of 256 pairs counted from the ROW START); segments are 128-byte but not
//                     4-KiB aligned in memory (row pitch 131584 B = 32.125 x 4 KiB)
//   V3 rowseg+4K    : as V2, but segments counted from 4-KiB-aligned ABSOLUTE addresses (the first / last workgroup of a row masked)
//   V4 linear half  : linear indexing over the first 8192 columns of each row only (a gap of half a row between row pieces)
//   V5 rowseg x4    : V2 with 1024 pairs per workgroup (4 stores per thread, 4 KiB apart)
// hipcc --offload-arch=gfx950 -O3 scripts/store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
static const int LD = 16448, NY = 16387, PR = LD / 2;       // pairs per row

__global__ __launch_bounds__(256) void v0(double *f, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) __builtin_nontemporal_store(d2{1.0, 1.0}, (d2 *)f + i);
}
__global__ __launch_bounds__(256) void v1(double *f, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const int c = (int)(i % PR);                            // pair index in its row
    if (c == 0) f[2 * i + 1] = 1.0;                         // column 0 not stored
    else if (c == PR - 1) f[2 * i] = 1.0;                   // column 16447 not stored
    else __builtin_nontemporal_store(d2{1.0, 1.0}, (d2 *)f + i);
}
template <int K>
__global__ __launch_bounds__(256) void v2(double *f, int segs)
{
    const int row = blockIdx.x / segs, sg = blockIdx.x - row * segs;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int c = sg * 256 * K + k * 256 + threadIdx.x;
        if (c >= PR) continue;
        const size_t i = (size_t)row * PR + c;
        if (c == 0) f[2 * i + 1] = 1.0;
        else if (c == PR - 1) f[2 * i] = 1.0;
        else __builtin_nontemporal_store(d2{1.0, 1.0}, (d2 *)f + i);
    }
}
__global__ __launch_bounds__(256) void v3(double *f, int segs)
{
    const int row = blockIdx.x / segs, sg = blockIdx.x - row * segs;
    const size_t row0 = (size_t)row * PR, first = row0 & ~(size_t)255;       // 256 pairs = 4 KiB
    const size_t i = first + (size_t)sg * 256 + threadIdx.x;
    if (i < row0 || i >= row0 + PR) return;
    const int c = (int)(i - row0);
    if (c == 0) f[2 * i + 1] = 1.0;
    else if (c == PR - 1) f[2 * i] = 1.0;
    else __builtin_nontemporal_store(d2{1.0, 1.0}, (d2 *)f + i);
}
__global__ __launch_bounds__(256) void v4(double *f, size_t n2half)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n2half) return;
    const size_t row = t / 4096, c = t % 4096;              // 4096 pairs = 8192 columns
    __builtin_nontemporal_store(d2{1.0, 1.0}, (d2 *)f + row * PR + c);
}

int main()
{
    const size_t n = (size_t)LD * NY, n2 = n / 2;
    double *f;
    CK(hipMalloc(&f, n * 8 + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            launch();
            CK(hipEventRecord(e0));
            for (int k = 0; k < 20; k++) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms / 20 < best) best = ms / 20;
        }
        printf("%-16s %8.4f ms  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, best, bytes / best / 1e6, bytes / best / 1e6 / 80);
    };
    const int segs1 = (PR + 255) / 256, segs4 = (PR + 1023) / 1024, segs3 = (PR + 255) / 256 + 1;
    timeit("V0 linear", n * 8.0, [&] { hipLaunchKernelGGL(v0, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, f, n2); });
    timeit("V1 linear+mask", (double)NY * (LD - 2) * 8.0, [&] { hipLaunchKernelGGL(v1, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, f, n2); });
    timeit("V2 rowseg", (double)NY * (LD - 2) * 8.0, [&] { hipLaunchKernelGGL(v2<1>, dim3((unsigned)((size_t)segs1 * NY)), dim3(256), 0, 0, f, segs1); });
    timeit("V3 rowseg+4K", (double)NY * (LD - 2) * 8.0, [&] { hipLaunchKernelGGL(v3, dim3((unsigned)((size_t)segs3 * NY)), dim3(256), 0, 0, f, segs3); });
    timeit("V4 linear half", (double)NY * 8192 * 8.0, [&] { hipLaunchKernelGGL(v4, dim3((unsigned)(((size_t)NY * 4096 + 255) / 256)), dim3(256), 0, 0, f, (size_t)NY * 4096); });
    timeit("V5 rowseg x4", (double)NY * (LD - 2) * 8.0, [&] { hipLaunchKernelGGL(v2<4>, dim3((unsigned)((size_t)segs4 * NY)), dim3(256), 0, 0, f, segs4); });
    timeit("V0 linear again", n * 8.0, [&] { hipLaunchKernelGGL(v0, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, f, n2); });
    return 0;
}
