// Streaming-bandwidth micro-benchmarks on one MI355X: what does a read+write stream of the
// Jacobi kernel's volume reach with different access shapes?  (known-good ceilings measured on
// the same hardware, guide rule 10).   hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// one 16-B element per thread, linear
__global__ void copy_linear(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <bool NT>
__global__ void copy_stride(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d2 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i);
        else b[i] = v;
    }
}
// each thread moves K consecutive-by-block elements per iteration (K loads in flight)
template <int K>
__global__ void copy_unroll(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n)
{
    size_t base = (size_t)blockIdx.x * blockDim.x * K + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * K;
    for (; base + (size_t)(K - 1) * blockDim.x < n; base += stride) {
        d2 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = a[base + (size_t)k * blockDim.x];
#pragma unroll
        for (int k = 0; k < K; k++) b[base + (size_t)k * blockDim.x] = v[k];
    }
}
// 2-D "march" copy with the Jacobi kernel's shape: block = 256 lanes x 16 B = 4 KiB of a row,
// walks `rows` rows of pitch ld; grid = nxb x nstrips (x fastest)
template <int U>
__global__ void copy_march(const d2 *__restrict__ a, d2 *__restrict__ b, int ld2, int nxb, int rows, int ny)
{
    const int bx = blockIdx.x % nxb, by = blockIdx.x / nxb;
    const int c = bx * 256 + threadIdx.x;
    if (c >= ld2) return;
    int j0 = by * rows, j1 = j0 + rows;
    if (j1 > ny) j1 = ny;
    int j = j0;
    for (; j + U <= j1; j += U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[(size_t)(j + u) * ld2 + c];
#pragma unroll
        for (int u = 0; u < U; u++) b[(size_t)(j + u) * ld2 + c] = v[u];
    }
    for (; j < j1; j++) b[(size_t)j * ld2 + c] = a[(size_t)j * ld2 + c];
}
__global__ void read_only(const d2 *__restrict__ a, double *__restrict__ out, size_t n)
{
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d2 v = a[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ void write_only(d2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = d2{1.0, 2.0};
}

template <typename F>
static double time_ms(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; k++) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / reps);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const int ld = 16448, ny = 16387;             // the bench field: 2.156 GB
    const size_t n = (size_t)ld * ny / 2;         // d2 elements
    const double gb = 2.0 * n * 16 / 1e9;         // read + write
    d2 *a, *b;
    double *out;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, n * 16));
    CK(hipMemset(b, 0, n * 16));
    auto rep = [&](const char *name, double ms, double gbytes) {
        printf("%-44s %8.4f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms, gbytes / ms * 1e3, gbytes / ms * 1e3 / 80);
        fflush(stdout);
    };
    rep("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); }), gb);
    rep("copy_linear (1 x 16B / thread)", time_ms([&] { copy_linear<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }), gb);
    for (int g : {1024, 2048, 4096, 8192, 16384})
        for (int nt = 0; nt < 2; nt++) {
            char nm[96];
            snprintf(nm, sizeof nm, "copy_stride grid=%d%s", g, nt ? " nt" : "");
            rep(nm, time_ms([&] {
                    if (nt) copy_stride<true><<<g, 256>>>(a, b, n);
                    else copy_stride<false><<<g, 256>>>(a, b, n);
                }), gb);
        }
    for (int g : {1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy_unroll<4> grid=%d", g);
        rep(nm, time_ms([&] { copy_unroll<4><<<g, 256>>>(a, b, n); }), gb);
        snprintf(nm, sizeof nm, "copy_unroll<8> grid=%d", g);
        rep(nm, time_ms([&] { copy_unroll<8><<<g, 256>>>(a, b, n); }), gb);
    }
    const int ld2 = ld / 2, nxb = (ld2 + 255) / 256;
    for (int rows : {1, 2, 4, 8, 16, 64, 265, 1024})
        for (int u : {4, 8}) {
            char nm[96];
            snprintf(nm, sizeof nm, "copy_march rows=%d U=%d (%d blocks)", rows, u, nxb * ((ny + rows - 1) / rows));
            const int ns = (ny + rows - 1) / rows;
            rep(nm, time_ms([&] {
                    if (u == 4) copy_march<4><<<nxb * ns, 256>>>(a, b, ld2, nxb, rows, ny);
                    else copy_march<8><<<nxb * ns, 256>>>(a, b, ld2, nxb, rows, ny);
                }), gb);
        }
    rep("read_only grid=4096", time_ms([&] { read_only<<<4096, 256>>>(a, out, n); }), gb / 2);
    rep("write_only grid=4096", time_ms([&] { write_only<<<4096, 256>>>(b, n); }), gb / 2);
    return 0;
}
