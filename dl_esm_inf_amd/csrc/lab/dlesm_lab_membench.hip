// libdlesm_lab.so -- MEASUREMENT TOOLING, not the product: nothing in libdlesm_hip.so or the Fortran layer uses it, and it
// shares no state with them (raw device pointers and a stream in, an error code out), so bench.py, scripts/ and tests/ load
// it NEXT TO the product library (include/dlesm_lab.h; Python: dl_esm_inf_amd._cabi.lab()).
//
// Measured ceilings for the roofline lines: a linear sweep that moves the same bytes as a kernel does --
// NR arrays read once, NW arrays written once, nothing else -- so that "fraction of what this many
// concurrent streams can reach on this box" is timed in the same process as the kernel itself
// (bench.py's `copy_ceiling` object).  One 16-byte element per thread per array and workgroups sweeping
// memory front to back: the shape that reaches the highest rate measured on this hardware
// (scripts/membench.hip, scripts/membench9.hip).  No reference counterpart.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "dlesm_lab.h"

namespace dlesm {

namespace {
thread_local char g_lab_err[256] = "";
int lab_fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int lab_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lab_err, sizeof g_lab_err, fmt, ap);
    va_end(ap);
    return -1;
}
#define LAB_REQUIRE(cond, ...) do { if (!(cond)) return lab_fail(__VA_ARGS__); } while (0)
} // namespace

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
struct StreamPtrs { const d2 *r[8]; d2 *w[6]; };   // (no restrict: written arrays may BE read arrays -- the in-place sweeps)

// NT bit 0: the second half of the read arrays is loaded non-temporally (the once-read old time level of the
// shallow-water step); bit 1: every store is non-temporal
template <int NR, int NW, int NT>
__global__ __launch_bounds__(256) void stream_copy_k(StreamPtrs p, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    d2 v[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) v[k] = ((NT & 1) && k >= (NR + 1) / 2) ? __builtin_nontemporal_load(p.r[k] + i) : p.r[k][i];
    d2 s = v[0];
#pragma unroll
    for (int k = 1; k < NR; k++) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const d2 o = NR == 1 ? v[0] : s + v[k % NR];
        if (NT & 2) __builtin_nontemporal_store(o, p.w[k] + i);
        else p.w[k][i] = o;
    }
}

template <int NR, int NW>
void launch_stream(const StreamPtrs &p, size_t n2, int nt, hipStream_t s)
{
    const unsigned grid = (unsigned)((n2 + 255) / 256);
    switch (nt & 3) {
    case 1: hipLaunchKernelGGL((stream_copy_k<NR, NW, 1>), dim3(grid), dim3(256), 0, s, p, n2); break;
    case 2: hipLaunchKernelGGL((stream_copy_k<NR, NW, 2>), dim3(grid), dim3(256), 0, s, p, n2); break;
    case 3: hipLaunchKernelGGL((stream_copy_k<NR, NW, 3>), dim3(grid), dim3(256), 0, s, p, n2); break;
    default: hipLaunchKernelGGL((stream_copy_k<NR, NW, 0>), dim3(grid), dim3(256), 0, s, p, n2); break;
    }
}

} // namespace

} // namespace dlesm

using namespace dlesm;

extern "C" const char *dlesm_lab_last_error(void) { return g_lab_err; }

extern "C" int dlesm_lab_stream_copy_f64(int nread, int nwrite, const double *const *src, double *const *dst, size_t n,
                                         int nt, void *stream)
{
    LAB_REQUIRE(src != nullptr && dst != nullptr, "dlesm_lab_stream_copy_f64: null pointer");
    LAB_REQUIRE(n % 2 == 0 && n / 2 < ((size_t)1 << 31) * 256, "dlesm_lab_stream_copy_f64: n = %zu must be even and below 2^40", n);
    StreamPtrs p{};
    LAB_REQUIRE(nread >= 1 && nread <= 8 && nwrite >= 1 && nwrite <= 6, "dlesm_lab_stream_copy_f64: %d read / %d written arrays",
                  nread, nwrite);
    for (int k = 0; k < nread; k++) {
        LAB_REQUIRE(src[k] != nullptr && (uintptr_t)src[k] % 16 == 0, "dlesm_lab_stream_copy_f64: read array %d null or not 16-byte aligned", k);
        p.r[k] = (const d2 *)src[k];
    }
    for (int k = 0; k < nwrite; k++) {
        LAB_REQUIRE(dst[k] != nullptr && (uintptr_t)dst[k] % 16 == 0, "dlesm_lab_stream_copy_f64: written array %d null or not 16-byte aligned", k);
        p.w[k] = (d2 *)dst[k];
    }
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t n2 = n / 2;
    if (nread == 1 && nwrite == 1) launch_stream<1, 1>(p, n2, nt, s);
    else if (nread == 2 && nwrite == 1) launch_stream<2, 1>(p, n2, nt, s);
    else if (nread == 3 && nwrite == 1) launch_stream<3, 1>(p, n2, nt, s);
    else if (nread == 4 && nwrite == 1) launch_stream<4, 1>(p, n2, nt, s);
    else if (nread == 6 && nwrite == 3) launch_stream<6, 3>(p, n2, nt, s);
    else if (nread == 6 && nwrite == 6) launch_stream<6, 6>(p, n2, nt, s);   // the filtered step: dst[3..5] may be src[3..5] (in place)
    else if (nread == 8 && nwrite == 1) launch_stream<8, 1>(p, n2, nt, s);
    else return lab_fail("dlesm_lab_stream_copy_f64: no %d-read / %d-write sweep (1+1, 2+1, 3+1, 4+1, 6+3, 6+6, 8+1)", nread, nwrite);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lab_fail("dlesm_lab_stream_copy_f64: launch failed: %s", hipGetErrorString(e));
    return 0;
}
