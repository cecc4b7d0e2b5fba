// What can NINE concurrent streams reach?  The shallow-water step reads six arrays and writes
// three (72 B/cell); the 6.3 TB/s copy ceiling of membench.hip is for one read + one write stream.
// This measures the same linear sweep with NR read arrays and NW written arrays of the 8192^2 field
// shape, one 16-B element per thread per array (the shape that reaches the copy ceiling), plus the
// 2-rows-per-thread form of the wave tiles and a form that re-reads two extra rows of three arrays
// (the u, v, p halo rows of shallow_tile<2>).     hipcc --offload-arch=gfx950 -O3 membench9.hip
#include <hip/hip_runtime.h>

#include <algorithm>
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
struct Ptrs { const d2 *r[6]; d2 *w[3]; };

template <int NR, int NW>
__global__ void stream_nm(Ptrs p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d2 v[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) v[k] = p.r[k][i];
    d2 s = v[0];
#pragma unroll
    for (int k = 1; k < NR; k++) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; k++) p.w[k][i] = s + v[k % NR];
}

// the same bytes as stream_nm<6,3>, but every workgroup touches only ONE triple (two read arrays, one
// written array): workgroup b works on triple (b / GRAN) % 3 -- three interleaved 2r+1w streams
template <int GRAN>
__global__ void stream_triples(Ptrs p, size_t n)
{
    const unsigned b = blockIdx.x;
    const unsigned k = (b / GRAN) % 3, chunk = (b / (3 * GRAN)) * GRAN + b % GRAN;
    const size_t i = (size_t)chunk * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d2 a = p.r[k][i], c = p.r[k + 3][i];
    p.w[k][i] = a + c;
}
// two 16-B elements per thread per array (fewer, fatter waves)
template <int NR, int NW, bool NT>
__global__ void stream_nm2(Ptrs p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x * 2 + threadIdx.x;
    if (i + blockDim.x >= n) return;
    d2 v[NR][2];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        if (NT && k >= 3) {
            v[k][0] = __builtin_nontemporal_load(p.r[k] + i);
            v[k][1] = __builtin_nontemporal_load(p.r[k] + i + blockDim.x);
        } else {
            v[k][0] = p.r[k][i];
            v[k][1] = p.r[k][i + blockDim.x];
        }
    }
    d2 s0 = v[0][0], s1 = v[0][1];
#pragma unroll
    for (int k = 1; k < NR; k++) { s0 += v[k][0]; s1 += v[k][1]; }
#pragma unroll
    for (int k = 0; k < NW; k++) {
        if (NT) {
            __builtin_nontemporal_store(s0 + v[k % NR][0], p.w[k] + i);
            __builtin_nontemporal_store(s1 + v[k % NR][1], p.w[k] + i + blockDim.x);
        } else {
            p.w[k][i] = s0 + v[k % NR][0];
            p.w[k][i + blockDim.x] = s1 + v[k % NR][1];
        }
    }
}

// wave-tile shape: a thread owns one 16-B column chunk of ROWS consecutive rows; tiles row-major
template <int NR, int NW, int ROWS, int HALO>
__global__ void stream_rows(Ptrs p, int ld2, int ny)
{
    const int tiles_x = (ld2 + blockDim.x - 1) / blockDim.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int c = tx * blockDim.x + threadIdx.x;
    const int j0 = ty * ROWS;
    if (c >= ld2 || j0 >= ny) return;
    d2 s = d2{0, 0};
    d2 v[NR][ROWS];
#pragma unroll
    for (int k = 0; k < NR; k++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            int j = j0 + r;
            if (j >= ny) j = ny - 1;
            v[k][r] = p.r[k][(size_t)j * ld2 + c];
        }
    if (HALO) { // rows j0-1 and j0+ROWS of the first three arrays
#pragma unroll
        for (int k = 0; k < 3 && k < NR; k++) {
            int ja = j0 > 0 ? j0 - 1 : 0, jb = j0 + ROWS < ny ? j0 + ROWS : ny - 1;
            s += p.r[k][(size_t)ja * ld2 + c] + p.r[k][(size_t)jb * ld2 + c];
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        d2 t = s;
#pragma unroll
        for (int k = 0; k < NR; k++) t += v[k][r];
        const int j = j0 + r;
        if (j < ny)
#pragma unroll
            for (int k = 0; k < NW; k++) p.w[k][(size_t)j * ld2 + c] = t + v[k % NR][r];
    }
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
    const int ld = 8256, ny = 8195;               // the shallow-water bench field: 541 MB each
    const size_t n = (size_t)ld * ny / 2;
    const int ld2 = ld / 2;
    Ptrs p, base;
    const size_t slack = 64u << 20;               // room to skew the arrays against each other
    for (int k = 0; k < 6; k++) { CK(hipMalloc((void **)&base.r[k], n * 16 + slack)); CK(hipMemset((void *)base.r[k], 0, n * 16 + slack)); }
    for (int k = 0; k < 3; k++) { CK(hipMalloc((void **)&base.w[k], n * 16 + slack)); CK(hipMemset((void *)base.w[k], 0, n * 16 + slack)); }
    p = base;
    for (int k = 0; k < 6; k++) printf("read  array %d at %p (mod 2 MiB: %zu KiB)\n", k, (void *)base.r[k], ((size_t)base.r[k] % (2u << 20)) >> 10);
    for (int k = 0; k < 3; k++) printf("write array %d at %p (mod 2 MiB: %zu KiB)\n", k, (void *)base.w[k], ((size_t)base.w[k] % (2u << 20)) >> 10);
    auto rep = [&](const char *name, double ms, int nr, int nw) {
        const double gb = (double)(nr + nw) * n * 16 / 1e9;
        printf("%-56s %8.4f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms, gb / ms * 1e3, gb / ms * 1e3 / 80);
        fflush(stdout);
    };
    const unsigned g = (unsigned)((n + 255) / 256);
    rep("linear 1 read + 1 write", time_ms([&] { stream_nm<1, 1><<<g, 256>>>(p, n); }), 1, 1);
    rep("linear 2 reads + 1 write", time_ms([&] { stream_nm<2, 1><<<g, 256>>>(p, n); }), 2, 1);
    rep("linear 3 reads + 3 writes", time_ms([&] { stream_nm<3, 3><<<g, 256>>>(p, n); }), 3, 3);
    rep("linear 6 reads + 3 writes", time_ms([&] { stream_nm<6, 3><<<g, 256>>>(p, n); }), 6, 3);
    rep("linear 6 reads + 1 write", time_ms([&] { stream_nm<6, 1><<<g, 256>>>(p, n); }), 6, 1);
    rep("linear 6 reads + 3 writes, 512 threads", time_ms([&] { stream_nm<6, 3><<<(unsigned)((n + 511) / 512), 512>>>(p, n); }), 6, 3);
    for (int threads : {256, 512}) {
        const int tx = (ld2 + threads - 1) / threads;
        char nm[128];
        snprintf(nm, sizeof nm, "tiles 2 rows, 6r+3w, %d threads", threads);
        rep(nm, time_ms([&] { stream_rows<6, 3, 2, 0><<<tx * ((ny + 1) / 2), threads>>>(p, ld2, ny); }), 6, 3);
        snprintf(nm, sizeof nm, "tiles 2 rows, 6r+3w + halo rows of 3 arrays, %d thr", threads);
        rep(nm, time_ms([&] { stream_rows<6, 3, 2, 1><<<tx * ((ny + 1) / 2), threads>>>(p, ld2, ny); }), 6, 3);
        snprintf(nm, sizeof nm, "tiles 4 rows, 6r+3w + halo rows of 3 arrays, %d thr", threads);
        rep(nm, time_ms([&] { stream_rows<6, 3, 4, 1><<<tx * ((ny + 3) / 4), threads>>>(p, ld2, ny); }), 6, 3);
        snprintf(nm, sizeof nm, "tiles 1 row, 6r+3w + halo rows of 3 arrays, %d thr", threads);
        rep(nm, time_ms([&] { stream_rows<6, 3, 1, 1><<<tx * ny, threads>>>(p, ld2, ny); }), 6, 3);
    }
    rep("triples (2r+1w per workgroup), interleave 1 group", time_ms([&] { stream_triples<1><<<3 * g, 256>>>(p, n); }), 6, 3);
    rep("triples (2r+1w per workgroup), interleave 8 groups", time_ms([&] { stream_triples<8><<<3 * ((g + 7) / 8 * 8), 256>>>(p, n); }), 6, 3);
    rep("triples (2r+1w per workgroup), interleave 64 groups", time_ms([&] { stream_triples<64><<<3 * ((g + 63) / 64 * 64), 256>>>(p, n); }), 6, 3);
    rep("triples (2r+1w per workgroup), interleave 1024 groups", time_ms([&] { stream_triples<1024><<<3 * ((g + 1023) / 1024 * 1024), 256>>>(p, n); }), 6, 3);
    rep("linear 6r+3w, 2 x 16 B per thread per array", time_ms([&] { stream_nm2<6, 3, false><<<(g + 1) / 2, 256>>>(p, n); }), 6, 3);
    rep("linear 6r+3w, 2 x 16 B, nt loads of 3 arrays + nt stores", time_ms([&] { stream_nm2<6, 3, true><<<(g + 1) / 2, 256>>>(p, n); }), 6, 3);
    // the same arrays skewed against each other: array k starts k * S bytes further on
    for (size_t S : {(size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)(1u << 20), (size_t)((4u << 20) + 8192 + 256)}) {
        for (int k = 0; k < 6; k++) p.r[k] = (const d2 *)((const char *)base.r[k] + k * S);
        for (int k = 0; k < 3; k++) p.w[k] = (d2 *)((char *)base.w[k] + (6 + k) * S);
        char nm[128];
        snprintf(nm, sizeof nm, "linear 6r+3w, array k skewed by k x %zu B", S);
        rep(nm, time_ms([&] { stream_nm<6, 3><<<g, 256>>>(p, n); }), 6, 3);
        snprintf(nm, sizeof nm, "linear 6r+1w, array k skewed by k x %zu B", S);
        rep(nm, time_ms([&] { stream_nm<6, 1><<<g, 256>>>(p, n); }), 6, 1);
    }
    return 0;
}
