// A PSy-layer loop nest whose kernel reads fields on all three C-grid point types AND a
// double-precision grid property: the free-surface (continuity) update of a NEMOLite2D-class model,
// kernel metadata
//     go_arg(GO_WRITE, GO_CT, GO_POINTWISE),                   ! ssha
//     go_arg(GO_READ,  GO_CT, GO_POINTWISE),                   ! sshn_t
//     go_arg(GO_READ,  GO_CU, GO_STENCIL(000,110,000)), ...    ! sshn_u, hu, un: (ji, jj) and (ji-1, jj)
//     go_arg(GO_READ,  GO_CV, GO_STENCIL(000,010,010)), ...    ! sshn_v, hv, vn: (ji, jj) and (ji, jj-1)
//     go_arg(GO_READ,  GO_R_SCALAR, GO_POINTWISE),             ! rdt
//     go_arg(GO_READ,  GO_GRID_AREA_T)                         ! grid%area_t   (argument_mod.f90:75-112)
// The PSy layer passes grid%area_t -- on the device its mirror grid%area_t_device (grid_mod.f90:104-150).
// The reference holds no such loop; the specification is frozen in DESIGN.md
// section 5.10:
//     r1 = (sshn_u(ji  ,jj) + hu(ji  ,jj)) * un(ji  ,jj)
//     r2 = (sshn_u(ji-1,jj) + hu(ji-1,jj)) * un(ji-1,jj)
//     r3 = (sshn_v(ji,jj  ) + hv(ji,jj  )) * vn(ji,jj  )
//     r4 = (sshn_v(ji,jj-1) + hv(ji,jj-1)) * vn(ji,jj-1)
//     ssha(ji,jj) = sshn_t(ji,jj) + (((r2 - r1) + r4) - r3) * rdt / area_t(ji,jj)
// every operation rounded, evaluated left to right as the Fortran expression is.
//
// 72 B/cell of algorithmic traffic (eight arrays read once, one written).  Wave tiles of 64 lanes x
// 2 columns x 2 rows swept linearly, like the other sweeps: the u-point operands at ji-1 come from the
// neighbouring lane (lane 0 fetches the one column outside the wave), the v-point operands at jj-1
// from the row loaded for the tile's row below.
#include "dlesm_internal.h"

namespace dlesm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double cont_point(double rdt, double st, double su, double su_w, double sv, double sv_s,
                                             double hu, double hu_w, double hv, double hv_s, double un, double un_w,
                                             double vn, double vn_s, double area)
{
    const double r1 = (su + hu) * un, r2 = (su_w + hu_w) * un_w;
    const double r3 = (sv + hv) * vn, r4 = (sv_s + hv_s) * vn_s;
    return st + (((r2 - r1) + r4) - r3) * rdt / area;
}

struct ContFields {
    const double *sshn_t, *sshn_u, *sshn_v, *hu, *hv, *un, *vn, *area_t;
    double *ssha;
};

constexpr int R = 2;

template <bool NT> __device__ __forceinline__ d2 ldv(const double *p)
{
    return NT ? __builtin_nontemporal_load((const d2 *)p) : *(const d2 *)p;
}

// NTM bit 0: the arrays read exactly once (T- and U-point fields, area_t) are loaded non-temporally;
// bit 1: ssha is stored non-temporally.  The V-point rows are re-read by the tile above: default policy.
template <int NTM>
__global__ __launch_bounds__(1024) void continuity_tile(ContFields f, double rdt, int ld, int x0, int x1, int y0, int y1,
                                                       int c_first, int nxw)
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
    const int ecol = (lane == 0 && m0) ? c * 2 - 1 : -1; // the west column this wave cannot get from a lane

    d2 sv[R + 1], hv[R + 1], vn[R + 1];                  // rows jb-1 .. je
#pragma unroll
    for (int k = 0; k < R + 1; k++) {
        int jj = jb - 1 + k;
        if (jj > je) jj = je;
        const size_t o = (size_t)jj * ld + (size_t)cl * 2;
        sv[k] = *(const d2 *)(f.sshn_v + o);
        hv[k] = *(const d2 *)(f.hv + o);
        vn[k] = *(const d2 *)(f.vn + o);
    }
    // every load of the tile first (rows beyond je clamped: loaded again, never used), the west column of lane 0 among
    // them -- an edge load behind a branch in the row loop is one more dependent round trip per row
    constexpr bool N1 = (NTM & 1) != 0;
    d2 st[R], ar[R], su[R], hu[R], un[R];
    double esu[R], ehu[R], eun[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        int jj = jb + k;
        if (jj > je) jj = je;
        const size_t row = (size_t)jj * ld, o = row + (size_t)cl * 2;
        st[k] = ldv<N1>(f.sshn_t + o), ar[k] = ldv<N1>(f.area_t + o);
        su[k] = ldv<N1>(f.sshn_u + o), hu[k] = ldv<N1>(f.hu + o), un[k] = ldv<N1>(f.un + o);
        const size_t eo = ecol >= 0 ? row + ecol : o;        // other lanes: a cell they have just loaded
        esu[k] = f.sshn_u[eo], ehu[k] = f.hu[eo], eun[k] = f.un[eo];
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int jj = jb + k;
        if (jj > je) break;
        const size_t row = (size_t)jj * ld;
        double su_w = from_lower<true>(su[k].y), hu_w = from_lower<true>(hu[k].y), un_w = from_lower<true>(un[k].y);
        if (ecol >= 0) { su_w = esu[k]; hu_w = ehu[k]; un_w = eun[k]; }
        const double o0 = cont_point(rdt, st[k].x, su[k].x, su_w, sv[k + 1].x, sv[k].x, hu[k].x, hu_w, hv[k + 1].x, hv[k].x,
                                     un[k].x, un_w, vn[k + 1].x, vn[k].x, ar[k].x);
        const double o1 = cont_point(rdt, st[k].y, su[k].y, su[k].x, sv[k + 1].y, sv[k].y, hu[k].y, hu[k].x, hv[k + 1].y,
                                     hv[k].y, un[k].y, un[k].x, vn[k + 1].y, vn[k].y, ar[k].y);
        double *po = f.ssha + row + (size_t)c * 2;
        if (m0 && m1) {
            if (NTM & 2) __builtin_nontemporal_store(d2{o0, o1}, (d2 *)po);
            else *(d2 *)po = d2{o0, o1};
        } else {
            if (m0) po[0] = o0;
            if (m1) po[1] = o1;
        }
    }
}

// one cell per thread: odd leading dimensions and unaligned bases
__global__ __launch_bounds__(256) void continuity_direct(ContFields f, double rdt, int ld, int x0, int x1, int y0, int y1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > x1) return;
    for (int j = y0 + blockIdx.y; j <= y1; j += gridDim.y) {
        const size_t o = (size_t)j * ld + i;
        f.ssha[o] = cont_point(rdt, f.sshn_t[o], f.sshn_u[o], f.sshn_u[o - 1], f.sshn_v[o], f.sshn_v[o - ld], f.hu[o],
                               f.hu[o - 1], f.hv[o], f.hv[o - ld], f.un[o], f.un[o - 1], f.vn[o], f.vn[o - ld], f.area_t[o]);
    }
}

} // namespace

} // namespace dlesm

using namespace dlesm;

extern "C" int dlesm_continuity_f64(double rdt, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                                    const double *sshn_t, const double *sshn_u, const double *sshn_v, const double *hu,
                                    const double *hv, const double *un, const double *vn, const double *area_t,
                                    double *ssha, void *stream)
{
    if (int rc = ensure_device()) return rc;
    if (xstop < xstart || ystop < ystart) return DLESM_OK;   // empty box: a zero-trip loop nest
    if (int rc = check_box("dlesm_continuity_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(sshn_t && sshn_u && sshn_v && hu && hv && un && vn && area_t && ssha, "continuity: null pointer");
    const ContFields f{sshn_t, sshn_u, sshn_v, hu, hv, un, vn, area_t, ssha};
    bool aligned = ld % 2 == 0;
    for (const double *q : {sshn_t, sshn_u, sshn_v, hu, hv, un, vn, area_t, (const double *)ssha})
        aligned = aligned && (uintptr_t)q % 16 == 0;
    DLESM_REQUIRE(ssha != sshn_u && ssha != sshn_v && ssha != hu && ssha != hv && ssha != un && ssha != vn,
                  "continuity: ssha aliases an input that is read at a neighbouring point");
    hipStream_t s = (hipStream_t)stream;
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    if (aligned && tuning("cont_kernel", 0) == 0) {
        const int c_first = (x0 / 2) & ~7, c_last = x1 / 2;  // tiles anchored on 128-byte lines of the row
        int nxw = (c_last - c_first + 64) / 64, tpb = 4;
        shape_for_tile_sweep(ld, x0, x1, y0, y1, &nxw, &tpb);
        const int strips = (y1 - y0 + R) / R;
        const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
        switch (tuning("cont_nt", 2) & 3) {
        case 1: hipLaunchKernelGGL(continuity_tile<1>, dim3(grid), dim3(64 * tpb), 0, s, f, rdt, ld, x0, x1, y0, y1, c_first, nxw); break;
        case 2: hipLaunchKernelGGL(continuity_tile<2>, dim3(grid), dim3(64 * tpb), 0, s, f, rdt, ld, x0, x1, y0, y1, c_first, nxw); break;
        case 3: hipLaunchKernelGGL(continuity_tile<3>, dim3(grid), dim3(64 * tpb), 0, s, f, rdt, ld, x0, x1, y0, y1, c_first, nxw); break;
        default: hipLaunchKernelGGL(continuity_tile<0>, dim3(grid), dim3(64 * tpb), 0, s, f, rdt, ld, x0, x1, y0, y1, c_first, nxw); break;
        }
    } else {
        const int h = y1 - y0 + 1;
        hipLaunchKernelGGL(continuity_direct, dim3((x1 - x0 + 256) / 256, h > 4096 ? 4096 : h), dim3(256), 0, s, f, rdt, ld,
                           x0, x1, y0, y1);
    }
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}
