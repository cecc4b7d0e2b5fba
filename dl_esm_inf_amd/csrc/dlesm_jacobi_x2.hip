// Two Jacobi-5 time steps in one sweep (temporal blocking).
//
//   t(i,j)   = J(in)(i,j)   for (i,j) in the intermediate box E,   in(i,j) elsewhere
//   out(i,j) = J(t)(i,j)    for (i,j) in the output box B
//
// with J(f)(i,j) = 0.25*((f(i-1,j)+f(i+1,j)) + (f(i,j-1)+f(i,j+1))), i.e. exactly two
// dlesm_stencil5_f64 calls through a ping-pong buffer whose cells outside E equal `in`'s
// (one GPU: E = B, the fixed boundary ring; distributed: E = B grown by one cell towards
// every neighbouring tile, `in` carrying depth-2 halos).  The intermediate never touches
// memory: 16 B per cell for TWO steps instead of 32.  Same expression tree per step as the
// single-step kernel, so the result is bit-identical to two single steps.
//
// Wave tile: 64 lanes x 2 doubles x (R+4) input rows -> (R+2) intermediate rows for all 64
// lanes -> R output rows for lanes 1..62; lanes 0 and 63 are halo lanes (their inner
// intermediate column is valid and feeds lane 1 / 62 by shuffle).  Row-major linear sweep
// and block-shape rule as in jacobi5_tile.
#include <algorithm>

#include "dlesm_internal.h"

namespace dlesm {

template <int R>
__global__ __launch_bounds__(1024) void jacobi5x2_tile(const double *__restrict__ in,
                                                      double *__restrict__ out, int ld, int ny, int x0,
                                                      int x1, int y0, int y1, int ex0, int ex1, int ey0,
                                                      int ey1, int cb, int nxw)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int xw = w % nxw;
    const int jb = y0 + (w / nxw) * R;
    if (jb > y1) return;
    int je = jb + R - 1;
    if (je > y1) je = y1;
    const int c = cb + xw * 62 - 1 + lane;
    if (c - lane + 1 > x1 / 2) return;                  // idle padding tile
    const int c_ld = ld / 2 - 1;
    const int cl = c < 0 ? 0 : (c > c_ld ? c_ld : c);
    const bool ol = lane >= 1 && lane <= 62 && c <= c_ld;
    const bool m0 = ol && 2 * c >= x0 && 2 * c <= x1;
    const bool m1 = ol && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;
    const bool e0 = 2 * c >= ex0 && 2 * c <= ex1;       // intermediate computed (else copied) here
    const bool e1 = 2 * c + 1 >= ex0 && 2 * c + 1 <= ex1;
    const double *pin = in + (size_t)cl * 2;
    d2 r[R + 4];
#pragma unroll
    for (int u = 0; u < R + 4; u++) {
        int jj = jb - 2 + u;
        if (jj > je + 2) jj = je + 2;                   // short last strip: loaded again, never used
        jj = jj < 0 ? 0 : (jj > ny - 1 ? ny - 1 : jj);  // rows outside the array feed discarded values only
        r[u] = *(const d2 *)(pin + (size_t)jj * ld);
    }
    d2 t[R + 2];
#pragma unroll
    for (int u = 0; u < R + 2; u++) {
        const int jt = jb - 1 + u;
        const double west = __shfl_up(r[u + 1].y, 1), east = __shfl_down(r[u + 1].x, 1);
        const double tx = 0.25 * ((west + r[u + 1].y) + (r[u].x + r[u + 2].x));
        const double ty = 0.25 * ((r[u + 1].x + east) + (r[u].y + r[u + 2].y));
        const bool rowin = jt >= ey0 && jt <= ey1;
        t[u].x = (rowin && e0) ? tx : r[u + 1].x;
        t[u].y = (rowin && e1) ? ty : r[u + 1].y;
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
        const double west = __shfl_up(t[u + 1].y, 1), east = __shfl_down(t[u + 1].x, 1);
        if (jb + u <= je) {
            const double o0 = 0.25 * ((west + t[u + 1].y) + (t[u].x + t[u + 2].x));
            const double o1 = 0.25 * ((t[u + 1].x + east) + (t[u].y + t[u + 2].y));
            double *po = out + (size_t)(jb + u) * ld + (size_t)c * 2;
            if (m0 && m1) *(d2 *)po = d2{o0, o1};
            else {
                if (m0) po[0] = o0;
                if (m1) po[1] = o1;
            }
        }
    }
}

// One cell per thread, neighbours through L1/L2: used when the arrays do not meet the 16-byte
// lane conditions of the tile kernel.  0-based inclusive boxes.
__global__ __launch_bounds__(256) void jacobi5x2_direct(const double *__restrict__ in,
                                                        double *__restrict__ out, int ld, int x0, int x1,
                                                        int y0, int y1, int ex0, int ex1, int ey0, int ey1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x, j = y0 + blockIdx.y;
    if (i > x1 || j > y1) return;
    auto T = [&](int ii, int jj) -> double {
        const size_t o = (size_t)jj * ld + ii;
        if (ii < ex0 || ii > ex1 || jj < ey0 || jj > ey1) return in[o];
        return 0.25 * ((in[o - 1] + in[o + 1]) + (in[o - ld] + in[o + ld]));
    };
    out[(size_t)j * ld + i] = 0.25 * ((T(i - 1, j) + T(i + 1, j)) + (T(i, j - 1) + T(i, j + 1)));
}

int launch_stencil5_x2(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart,
                       int ystop, int exstart, int exstop, int eystart, int eystop, hipStream_t s)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK; // empty box: a zero-trip loop nest
    if (int rc = check_box("dlesm_stencil5_x2_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && in != out, "stencil5_x2: null or aliased arrays");
    const bool empty_e = exstop < exstart || eystop < eystart;
    if (!empty_e)
        if (int rc = check_box("dlesm_stencil5_x2_f64 (intermediate box)", ld, ny, exstart, exstop, eystart,
                               eystop, 1))
            return rc;
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    int ex0 = exstart - 1, ex1 = exstop - 1, ey0 = eystart - 1, ey1 = eystop - 1;
    if (empty_e) { ex0 = ey0 = 1; ex1 = ey1 = 0; }
    // 16-byte lanes need the last column any lane may need (ex1 + 1, x1 + 1) inside the last
    // whole 2-column chunk of a row, and 16-byte aligned bases
    const int last = std::max(x1, ex1) + 1;
    const bool vec2 = !(tuning("j5_variant", 0) & 4) && last <= 2 * (ld / 2) - 1 && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    if (!vec2) {
        dim3 grid((unsigned)((x1 - x0 + 256) / 256), (unsigned)(y1 - y0 + 1));
        // grid.y is limited to 65535 rows per launch
        for (int yb = y0; yb <= y1; yb += 65535) {
            const int ye = std::min(y1, yb + 65534);
            grid.y = (unsigned)(ye - yb + 1);
            hipLaunchKernelGGL(jacobi5x2_direct, grid, dim3(256), 0, s, in, out, ld, x0, x1, yb, ye, ex0, ex1, ey0,
                               ey1);
        }
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    int R = tuning("j5x2_tile_rows", 4);
    if (R != 2 && R != 3 && R != 6 && R != 8) R = 4;
    const int cb = x0 / 2, c_last = x1 / 2;
    int nxw = (c_last - cb + 62) / 62, tpb = 4;
    choose_block_shape(&nxw, &tpb);
    const int strips = (y1 - y0 + R) / R;
    const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
#define DLESM_X2(RR)                                                                                        \
    hipLaunchKernelGGL(jacobi5x2_tile<RR>, dim3(grid), dim3(64 * tpb), 0, s, in, out, ld, ny, x0, x1, y0, y1, \
                       ex0, ex1, ey0, ey1, cb, nxw)
    switch (R) {
    case 2: DLESM_X2(2); break;
    case 3: DLESM_X2(3); break;
    case 6: DLESM_X2(6); break;
    case 8: DLESM_X2(8); break;
    default: DLESM_X2(4); break;
    }
#undef DLESM_X2
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

} // namespace dlesm

extern "C" int dlesm_stencil5_x2_f64(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                                     int ystart, int ystop, int exstart, int exstop, int eystart,
                                     int eystop, void *stream)
{
    dlesm::clear_error();
    if (int rc = dlesm::ensure_device()) return rc;
    return dlesm::launch_stencil5_x2(in, out, ld, ny, xstart, xstop, ystart, ystop, exstart, exstop, eystart,
                                     eystop, (hipStream_t)stream);
}
