// A PSy-layer loop nest whose kernel consumes a GRID PROPERTY: the
// land/sea mask at T points, requested by a kernel through GO_GRID_MASK_T in its metadata
// (argument_mod.f90:75-112) and passed by the PSy layer as grid%tmask -- here its device mirror
// grid%tmask_device (grid_mod.f90:104-106).  tmask(1:grid%nx, 1:grid%ny) is a default-integer
// array with the field layout (grid_mod.f90:394): 1 = wet, 0 = land, -1 = outside the domain
// (the NEMOLite2D convention the GOcean kernels use).
//
// Masked 5-point Jacobi (specification frozen in DESIGN.md section 5.7; the reference has no
// stencil loop): for (ji, jj) in the box
//     if (tmask(ji,jj) <= 0)  out(ji,jj) = in(ji,jj)                      ! dry: carried over
//     else  w = in(ji-1,jj) if tmask(ji-1,jj) > 0 else in(ji,jj)          ! dry neighbour: mirrored
//           e, s, n likewise                                              !   (no-flux coast)
//           out(ji,jj) = 0.25*((w+e)+(s+n))
//
// 20 B/cell of algorithmic traffic (8 B in + 4 B mask read, 8 B written).  Same linear wave-tile
// sweep as jacobi5_tile: 64 lanes x 2 columns x 2 rows, the four rows of `in` and of the mask a
// tile needs are loaded up front (16-byte and 8-byte lanes), west/east values and masks come from
// the neighbouring lane, lanes 0 and 63 fetch the one column outside the wave.
#include "dlesm_internal.h"

namespace dlesm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double masked_point(double c, double w, double e, double s, double n, int mc, int mw,
                                               int me, int ms, int mn)
{
    const double ww = mw > 0 ? w : c, ee = me > 0 ? e : c, ss = ms > 0 ? s : c, nn = mn > 0 ? n : c;
    const double r = 0.25 * ((ww + ee) + (ss + nn));
    return mc > 0 ? r : c;
}

constexpr int R = 2;

template <bool NTS>
__global__ __launch_bounds__(1024) void jacobi5_masked_tile(const double *__restrict__ in,
                                                           double *__restrict__ out,
                                                           const int *__restrict__ tmask, int ld, int x0,
                                                           int x1, int y0, int y1, int c_first, int nxw)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int xw = w % nxw, jb = y0 + (w / nxw) * R;
    if (jb > y1) return;
    const int je = jb + R - 1 > y1 ? y1 : jb + R - 1;
    const int c = c_first + xw * 64 + lane;              // this lane's chunk (2 columns)
    if (c - lane > x1 / 2) return;                       // idle padding tile
    const int c_last = x1 / 2, c_ld = ld / 2 - 1;
    const int cl = c < c_ld ? c : c_ld;
    const bool m0 = c <= c_last && c * 2 >= x0 && c * 2 <= x1;
    const bool m1 = c <= c_last && c * 2 + 1 >= x0 && c * 2 + 1 <= x1;
    int ecol = -1;                                       // the one column this wave cannot get from a lane
    if (lane == 0 && m0) ecol = c * 2 - 1;
    if (lane == 63 && m1) ecol = c * 2 + 2;

    d2 v[R + 2];
    i2 m[R + 2];
    double ev[R + 2];
    int em[R + 2];
#pragma unroll
    for (int k = 0; k < R + 2; k++) {
        int jj = jb - 1 + k;
        if (jj > je + 1) jj = je + 1;
        const size_t o = (size_t)jj * ld;
        v[k] = *(const d2 *)(in + o + (size_t)cl * 2);
        m[k] = *(const i2 *)(tmask + o + (size_t)cl * 2);
        ev[k] = ecol >= 0 ? in[o + ecol] : 0.0;
        em[k] = ecol >= 0 ? tmask[o + ecol] : 0;
    }
#pragma unroll
    for (int k = 1; k <= R; k++) {
        if (jb + k - 1 > je) break;
        // whole-wave shifts on the VALU (DPP), not through the LDS pipe: lanes 0 / 63 take the edge loads below
        double vw = from_lower<true>(v[k].y), ve = from_upper<true>(v[k].x);
        int mw = __builtin_amdgcn_mov_dpp(m[k].y, 0x138, 0xf, 0xf, true), me = __builtin_amdgcn_mov_dpp(m[k].x, 0x130, 0xf, 0xf, true);
        if (lane == 0) { vw = ev[k]; mw = em[k]; }
        if (lane == 63) { ve = ev[k]; me = em[k]; }
        const double o0 = masked_point(v[k].x, vw, v[k].y, v[k - 1].x, v[k + 1].x, m[k].x, mw, m[k].y, m[k - 1].x, m[k + 1].x);
        const double o1 = masked_point(v[k].y, v[k].x, ve, v[k - 1].y, v[k + 1].y, m[k].y, m[k].x, me, m[k - 1].y, m[k + 1].y);
        double *po = out + (size_t)(jb + k - 1) * ld + (size_t)c * 2;
        if (m0 && m1) {
            if (NTS) __builtin_nontemporal_store(d2{o0, o1}, (d2 *)po);
            else *(d2 *)po = d2{o0, o1};
        } else {
            if (m0) po[0] = o0;
            if (m1) po[1] = o1;
        }
    }
}

// one cell per thread: odd leading dimensions and unaligned bases
__global__ __launch_bounds__(256) void jacobi5_masked_direct(const double *__restrict__ in,
                                                            double *__restrict__ out,
                                                            const int *__restrict__ tmask, int ld, int x0,
                                                            int x1, int y0, int y1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > x1) return;
    for (int j = y0 + blockIdx.y; j <= y1; j += gridDim.y) {
        const size_t o = (size_t)j * ld + i;
        out[o] = masked_point(in[o], in[o - 1], in[o + 1], in[o - ld], in[o + ld], tmask[o], tmask[o - 1],
                              tmask[o + 1], tmask[o - ld], tmask[o + ld]);
    }
}

} // namespace

} // namespace dlesm

using namespace dlesm;

extern "C" int dlesm_stencil5_masked_f64(const double *in, double *out, const int *tmask, int ld, int ny,
                                         int xstart, int xstop, int ystart, int ystop, void *stream)
{
    if (int rc = ensure_device()) return rc;
    if (xstop < xstart || ystop < ystart) return DLESM_OK;   // empty box: a zero-trip loop nest
    if (int rc = check_box("dlesm_stencil5_masked_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && tmask != nullptr && in != out,
                  "masked stencil5: null or aliased arrays");
    hipStream_t s = (hipStream_t)stream;
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    const bool tile = ld % 2 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)tmask % 8 == 0 &&
                      tuning("j5m_kernel", 0) == 0;
    if (tile) {
        const int c_first = (x0 / 2) & ~7, c_last = x1 / 2;  // tiles anchored on 128-byte lines of the row
        int nxw = (c_last - c_first + 64) / 64, tpb = 4;
        shape_for_tile_sweep(ld, x0, x1, y0, y1, &nxw, &tpb);
        const int strips = (y1 - y0 + R) / R;
        const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
        if (nt_stores_for(ld, y0, y1))
            hipLaunchKernelGGL(jacobi5_masked_tile<true>, dim3(grid), dim3(64 * tpb), 0, s, in, out, tmask, ld, x0, x1, y0,
                               y1, c_first, nxw);
        else
            hipLaunchKernelGGL(jacobi5_masked_tile<false>, dim3(grid), dim3(64 * tpb), 0, s, in, out, tmask, ld, x0, x1, y0,
                               y1, c_first, nxw);
    } else {
        const int h = y1 - y0 + 1;
        hipLaunchKernelGGL(jacobi5_masked_direct, dim3((x1 - x0 + 256) / 256, h > 4096 ? 4096 : h), dim3(256), 0, s, in,
                           out, tmask, ld, x0, x1, y0, y1);
    }
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}
