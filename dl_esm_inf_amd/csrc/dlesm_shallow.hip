// Shallow-water u/v/h update as a register-tiled linear sweep.
//
// 72 B/cell of algorithmic traffic: u, v, p (3x3 footprint), uold, vold, pold read once,
// unew, vnew, pnew written once; cu, cv, z, h never touch memory.
//
// Work unit: a wave tile of 62 output lanes x 2 doubles x R rows.  A wave loads 64 lanes
// (one 16-byte-aligned, 1 KiB-contiguous access per row): lanes 0 and 63 are halo lanes whose
// values only feed their neighbours through wave64 shuffles, so there are no scattered edge
// loads and every cross-lane value -- raw (p, v east; u west) or derived (cu, z west; cv, h
// east) -- is one shuffle away.  Tiles are numbered row-major and workgroups sweep memory
// linearly in dispatch order, exactly as jacobi5_tile does (see the notes there); the
// (R+2)-row overlap between vertically adjacent tiles is served by L2 / Infinity Cache.
//
// The expression trees are exactly those of the specification in DESIGN.md section 6 and the
// file is compiled with -ffp-contract=off, so results agree bit for bit with the CPU checker
// used by the tests.
#include "dlesm_internal.h"

namespace dlesm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

struct V2 {
    double x, y;
};

__device__ __forceinline__ V2 ld2(const double *p)
{
    d2 t = *(const d2 *)p;
    return V2{t.x, t.y};
}
// value of the column to the east / west of each of the lane's two columns
// (DPP: whole-wave shift on the VALU instead of ds_bpermute, see dlesm_internal.h)
template <bool DPP> __device__ __forceinline__ V2 east_of(const V2 &a) { return V2{a.y, from_upper<DPP>(a.x)}; }
template <bool DPP> __device__ __forceinline__ V2 west_of(const V2 &a) { return V2{from_lower<DPP>(a.y), a.x}; }

#define EW(expr_x, expr_y) V2{(expr_x), (expr_y)}

template <int R, bool DPP>
__global__ __launch_bounds__(512) void shallow_tile(
    dlesm_sw_params q, int ld, int x0, int x1, int y0, int y1, int cb, int nxw,
    const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ p,
    const double *__restrict__ uold, const double *__restrict__ vold, const double *__restrict__ pold,
    double *__restrict__ unew, double *__restrict__ vnew, double *__restrict__ pnew)
{
    auto east = [](const V2 &a) { return east_of<DPP>(a); };
    auto west = [](const V2 &a) { return west_of<DPP>(a); };
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int xw = w % nxw, strip = w / nxw;
    const int jb = y0 + strip * R;
    if (jb > y1) return;
    int je = jb + R - 1;
    if (je > y1) je = y1;
    const int c = cb + xw * 62 - 1 + lane;             // this lane's chunk (2 columns)
    if (c - lane + 1 > x1 / 2) return;                 // idle padding tile
    const int c_ld = ld / 2 - 1;
    const int cl = c < 0 ? 0 : (c > c_ld ? c_ld : c);  // halo / trailing lanes: any valid chunk
    const bool out_lane = lane >= 1 && lane <= 62 && c <= c_ld;
    const bool m0 = out_lane && 2 * c >= x0 && 2 * c <= x1;
    const bool m1 = out_lane && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;

    const size_t col = (size_t)cl * 2;
    // rows jb-1 .. jb+R of u, v, p ; rows jb .. jb+R-1 of the old fields (clamped past je+1 / je)
    V2 U[R + 2], Vv[R + 2], P[R + 2], UO[R], VO[R], PO[R];
#pragma unroll
    for (int k = 0; k < R + 2; k++) {
        int jj = jb - 1 + k;
        if (jj > je + 1) jj = je + 1;
        const size_t o = (size_t)jj * ld + col;
        U[k] = ld2(u + o);
        Vv[k] = ld2(v + o);
        P[k] = ld2(p + o);
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        int jj = jb + k;
        if (jj > je) jj = je;
        const size_t o = (size_t)jj * ld + col;
        UO[k] = ld2(uold + o);
        VO[k] = ld2(vold + o);
        PO[k] = ld2(pold + o);
    }

    // raw neighbours
    V2 Pe[R + 2], Ve[R + 1], Uw[R + 2];
#pragma unroll
    for (int k = 0; k < R + 2; k++) Pe[k] = east(P[k]);
#pragma unroll
    for (int k = 0; k < R + 1; k++) Ve[k] = east(Vv[k]);
#pragma unroll
    for (int k = 1; k < R + 2; k++) Uw[k] = west(U[k]);

    // intermediates at the lane's own columns: index k = row jb-1+k
    V2 CU[R + 2], CV[R + 1], Z[R + 1], H[R + 2];
#pragma unroll
    for (int k = 1; k < R + 2; k++) {
        CU[k] = EW(0.5 * (Pe[k].x + P[k].x) * U[k].x, 0.5 * (Pe[k].y + P[k].y) * U[k].y);
        H[k] = EW(P[k].x + 0.25 * (U[k].x * U[k].x + Uw[k].x * Uw[k].x + Vv[k].x * Vv[k].x +
                                   Vv[k - 1].x * Vv[k - 1].x),
                  P[k].y + 0.25 * (U[k].y * U[k].y + Uw[k].y * Uw[k].y + Vv[k].y * Vv[k].y +
                                   Vv[k - 1].y * Vv[k - 1].y));
    }
#pragma unroll
    for (int k = 0; k < R + 1; k++) {
        CV[k] = EW(0.5 * (P[k + 1].x + P[k].x) * Vv[k].x, 0.5 * (P[k + 1].y + P[k].y) * Vv[k].y);
        Z[k] = EW((q.fsdx * (Ve[k].x - Vv[k].x) - q.fsdy * (U[k + 1].x - U[k].x)) /
                      (P[k].x + Pe[k].x + Pe[k + 1].x + P[k + 1].x),
                  (q.fsdx * (Ve[k].y - Vv[k].y) - q.fsdy * (U[k + 1].y - U[k].y)) /
                      (P[k].y + Pe[k].y + Pe[k + 1].y + P[k + 1].y));
    }
    // derived neighbours
    V2 CUw[R + 2], Zw[R + 1], CVe[R + 1], He[R + 1];
#pragma unroll
    for (int k = 1; k < R + 2; k++) CUw[k] = west(CU[k]);
#pragma unroll
    for (int k = 1; k < R + 1; k++) Zw[k] = west(Z[k]);
#pragma unroll
    for (int k = 0; k < R + 1; k++) CVe[k] = east(CV[k]);
#pragma unroll
    for (int k = 1; k < R + 1; k++) He[k] = east(H[k]);

#pragma unroll
    for (int k = 1; k <= R; k++) {
        const int jj = jb - 1 + k;
        if (jj > je) break;
        const V2 un = EW(UO[k - 1].x + q.tdts8 * (Z[k].x + Z[k - 1].x) *
                                           (CVe[k].x + CV[k].x + CV[k - 1].x + CVe[k - 1].x) -
                             q.tdtsdx * (He[k].x - H[k].x),
                         UO[k - 1].y + q.tdts8 * (Z[k].y + Z[k - 1].y) *
                                           (CVe[k].y + CV[k].y + CV[k - 1].y + CVe[k - 1].y) -
                             q.tdtsdx * (He[k].y - H[k].y));
        const V2 vn = EW(VO[k - 1].x - q.tdts8 * (Z[k].x + Zw[k].x) *
                                           (CU[k + 1].x + CUw[k + 1].x + CUw[k].x + CU[k].x) -
                             q.tdtsdy * (H[k + 1].x - H[k].x),
                         VO[k - 1].y - q.tdts8 * (Z[k].y + Zw[k].y) *
                                           (CU[k + 1].y + CUw[k + 1].y + CUw[k].y + CU[k].y) -
                             q.tdtsdy * (H[k + 1].y - H[k].y));
        const V2 pn = EW(PO[k - 1].x - q.tdtsdx * (CU[k].x - CUw[k].x) - q.tdtsdy * (CV[k].x - CV[k - 1].x),
                         PO[k - 1].y - q.tdtsdx * (CU[k].y - CUw[k].y) - q.tdtsdy * (CV[k].y - CV[k - 1].y));
        const size_t o = (size_t)jj * ld + (size_t)c * 2;
        if (m0 && m1) {
            *(d2 *)(unew + o) = d2{un.x, un.y};
            *(d2 *)(vnew + o) = d2{vn.x, vn.y};
            *(d2 *)(pnew + o) = d2{pn.x, pn.y};
        } else {
            if (m0) { unew[o] = un.x; vnew[o] = vn.x; pnew[o] = pn.x; }
            if (m1) { unew[o + 1] = un.y; vnew[o + 1] = vn.y; pnew[o + 1] = pn.y; }
        }
    }
}

} // namespace

void launch_shallow_tile(const dlesm_sw_params &q, int ld, int x0, int x1, int y0, int y1,
                         const double *u, const double *v, const double *p, const double *uold,
                         const double *vold, const double *pold, double *unew, double *vnew,
                         double *pnew, hipStream_t s)
{
    const int cb = x0 / 2, c_last = x1 / 2;              // first / last chunk holding an output column
    int nxw = (c_last - cb + 62) / 62, tpb = 4;          // 62 output chunks per wave tile
    // This kernel's landscape differs from the Jacobi one (nine arrays in flight): an exhaustive search at
    // 8192^2 (scripts/shallow_probe.py 8192 search; 67 tiles per row) finds 8 waves per group JUST ABOVE a
    // multiple of 8 groups best (67 tiles 0.844 ms, 68 0.846, 69 0.852), 4 waves at 23.75 groups equal
    // (0.846), and the region the Jacobi rule would pick -- 15.75 groups per row -- 25 % slower.  So: the
    // group size whose 8-group multiple lies closest below the row, no padding when the row is within
    // 3/8 group past it, else padding up to the next multiple (+1 tile when that lands on it exactly).
    if (!tuning("j5_autoshape", 1) || tuning("j5_tpb", 0) || nxw < 16) {
        choose_block_shape(&nxw, &tpb);                  // experiments and thin boxes: the shared path
    } else {
        double best = 1e9;
        int pad = 0;
        for (int cand : {8, 4, 2}) {
            const int period = 8 * cand, slack = 3 * cand / 8;
            if (nxw < period) continue;
            const int r = nxw % period, p = r <= slack ? 0 : period - r;
            const double cost = (double)p / nxw + (cand == 8 ? 0.0 : cand == 4 ? 0.01 : 0.03);
            if (cost < best) { best = cost; tpb = cand; pad = p; }
        }
        if (best > 0.25) { tpb = 4; pad = 0; }
        nxw += pad;
        if (nxw >= 128 && nxw % (8 * tpb) == 0) nxw += 1;
    }
    if (tpb > 8) tpb = 8;                                // the kernel is bounded to 512 threads
    int R = tuning("sw_tile_rows", 2);
    if (R != 1 && R != 3) R = 2;
    const int h = y1 - y0 + 1, strips = (h + R - 1) / R;
    const long tiles = (long)nxw * strips;
    const unsigned grid = (unsigned)((tiles + tpb - 1) / tpb);
    const bool dpp = tuning("sw_dpp", 1);
#define DLESM_SW(RR)                                                                                           \
    do {                                                                                                       \
        if (dpp)                                                                                               \
            hipLaunchKernelGGL((shallow_tile<RR, true>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, x0, x1, y0, y1, \
                               cb, nxw, u, v, p, uold, vold, pold, unew, vnew, pnew);                          \
        else                                                                                                   \
            hipLaunchKernelGGL((shallow_tile<RR, false>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, x0, x1, y0, y1, \
                               cb, nxw, u, v, p, uold, vold, pold, unew, vnew, pnew);                          \
    } while (0)
    if (R == 1) DLESM_SW(1);
    else if (R == 3) DLESM_SW(3);
    else DLESM_SW(2);
#undef DLESM_SW
}

} // namespace dlesm
