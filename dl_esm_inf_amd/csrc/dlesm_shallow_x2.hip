// TWO leapfrog steps of the shallow-water update (NE offset) per launch.
//
// The fused single step sits on the measured ceiling of its nine concurrent HBM streams (0.75 of peak = 1.03 x a plain
// 6-read + 3-write copy); the only way to more cells per second is fewer bytes per time step.  With levels n (u, v, p) and
// n-1 (uold, vold, pold) in memory, one launch of this kernel writes level n+1 AND level n+2:
//
//     level n+1 = step(level n,   level n-1)        on the box and, in registers only, one cell around each wave tile
//     level n+2 = step(level n+1, level n)          on the box
//
// six arrays read once, six written once: 96 B/cell per TWO steps = 48 B/cell/step against 72.  Each stage is the expression
// tree of the single step, operand for operand (the first stage evaluated redundantly on the one-cell rim of the tile, as
// temporal blocking always does), so the two levels are bit-identical to two calls of dlesm_shallow_step_f64.
//
// Wave tile: 62 (or 56, template parameter HL) output lanes x 2 doubles x R rows.  A lane holds two columns, so the ONE halo lane
// per side the single step needs already covers the two columns two steps need: stage 1 is valid on columns 1 .. 126 of the
// wave's 128, stage 2 on the 124 columns of lanes 1 .. 62.  Vertically the tile loads rows jb-2 .. jb+R+1 of level n and rows
// jb-1 .. jb+R of level n-1.  Cells of level n+1 that lie OUTSIDE the box (the fixed boundary ring of a non-periodic model)
// are not computed: they are what the level-n+1 arrays hold there, loaded by the tiles that touch the edge of the box only.
//
// Level n+2 cannot go into the arrays of level n-1 in place, as the filtered step's old level does: stage 1 reads level n-1
// one cell around the tile, i.e. cells a neighbouring tile would be overwriting.  Four time levels, twelve arrays.
#include <cstdint>

#include "dlesm_internal.h"
#include "dlesm_device.h"

namespace dlesm {

namespace {

typedef double x2_d2 __attribute__((ext_vector_type(2)));
struct W2 { double x, y; };

template <bool NT>
__device__ __forceinline__ W2 xld(const double *p)
{
    x2_d2 t = NT ? __builtin_nontemporal_load((const x2_d2 *)p) : *(const x2_d2 *)p;
    return W2{t.x, t.y};
}
template <bool NT>
__device__ __forceinline__ void xst(double *p, const W2 &v)
{
    if constexpr (NT) __builtin_nontemporal_store(x2_d2{v.x, v.y}, (x2_d2 *)p);
    else *(x2_d2 *)p = x2_d2{v.x, v.y};
}
__device__ __forceinline__ W2 east(const W2 &a) { return W2{a.y, from_upper<true>(a.x)}; }
__device__ __forceinline__ W2 west(const W2 &a) { return W2{from_lower<true>(a.y), a.x}; }
#define XW(ex, ey) W2{(ex), (ey)}

// One step on register rows: U, V, P hold NR + 2 rows (index k = row r0 - 1 + k), the old level NR rows (index k - 1 = row
// r0 - 1 + k, k = 1 .. NR); the new level comes out for those NR rows.  The expression trees of shallow_tile_body
// (dlesm_shallow.hip) = DESIGN.md section 6, operand for operand.
template <int NR>
__device__ __forceinline__ void sw_step_rows(const dlesm_sw_params &q, const W2 (&U)[NR + 2], const W2 (&V)[NR + 2], const W2 (&P)[NR + 2],
                                             const W2 (&UO)[NR], const W2 (&VO)[NR], const W2 (&PO)[NR], W2 (&UN)[NR], W2 (&VN)[NR],
                                             W2 (&PN)[NR])
{
    W2 Pe[NR + 2], Ve[NR + 1], Uw[NR + 2];
#pragma unroll
    for (int k = 0; k < NR + 2; k++) Pe[k] = east(P[k]);
#pragma unroll
    for (int k = 0; k < NR + 1; k++) Ve[k] = east(V[k]);
#pragma unroll
    for (int k = 1; k < NR + 2; k++) Uw[k] = west(U[k]);
    W2 CU[NR + 2], CV[NR + 1], Z[NR + 1], H[NR + 2];
#pragma unroll
    for (int k = 1; k < NR + 2; k++) {
        CU[k] = XW(0.5 * (Pe[k].x + P[k].x) * U[k].x, 0.5 * (Pe[k].y + P[k].y) * U[k].y);
        H[k] = XW(P[k].x + 0.25 * (U[k].x * U[k].x + Uw[k].x * Uw[k].x + V[k].x * V[k].x + V[k - 1].x * V[k - 1].x),
                  P[k].y + 0.25 * (U[k].y * U[k].y + Uw[k].y * Uw[k].y + V[k].y * V[k].y + V[k - 1].y * V[k - 1].y));
    }
#pragma unroll
    for (int k = 0; k < NR + 1; k++) {
        CV[k] = XW(0.5 * (P[k + 1].x + P[k].x) * V[k].x, 0.5 * (P[k + 1].y + P[k].y) * V[k].y);
        Z[k] = XW((q.fsdx * (Ve[k].x - V[k].x) - q.fsdy * (U[k + 1].x - U[k].x)) / (P[k].x + Pe[k].x + Pe[k + 1].x + P[k + 1].x),
                  (q.fsdx * (Ve[k].y - V[k].y) - q.fsdy * (U[k + 1].y - U[k].y)) / (P[k].y + Pe[k].y + Pe[k + 1].y + P[k + 1].y));
    }
    W2 CUw[NR + 2], Zw[NR + 1], CVe[NR + 1], He[NR + 1];
#pragma unroll
    for (int k = 1; k < NR + 2; k++) CUw[k] = west(CU[k]);
#pragma unroll
    for (int k = 1; k < NR + 1; k++) Zw[k] = west(Z[k]);
#pragma unroll
    for (int k = 0; k < NR + 1; k++) CVe[k] = east(CV[k]);
#pragma unroll
    for (int k = 1; k < NR + 1; k++) He[k] = east(H[k]);
#pragma unroll
    for (int k = 1; k <= NR; k++) {
        UN[k - 1] = XW(UO[k - 1].x + q.tdts8 * (Z[k].x + Z[k - 1].x) * (CVe[k].x + CV[k].x + CV[k - 1].x + CVe[k - 1].x) -
                           q.tdtsdx * (He[k].x - H[k].x),
                       UO[k - 1].y + q.tdts8 * (Z[k].y + Z[k - 1].y) * (CVe[k].y + CV[k].y + CV[k - 1].y + CVe[k - 1].y) -
                           q.tdtsdx * (He[k].y - H[k].y));
        VN[k - 1] = XW(VO[k - 1].x - q.tdts8 * (Z[k].x + Zw[k].x) * (CU[k + 1].x + CUw[k + 1].x + CUw[k].x + CU[k].x) -
                           q.tdtsdy * (H[k + 1].x - H[k].x),
                       VO[k - 1].y - q.tdts8 * (Z[k].y + Zw[k].y) * (CU[k + 1].y + CUw[k + 1].y + CUw[k].y + CU[k].y) -
                           q.tdtsdy * (H[k + 1].y - H[k].y));
        PN[k - 1] = XW(PO[k - 1].x - q.tdtsdx * (CU[k].x - CUw[k].x) - q.tdtsdy * (CV[k].x - CV[k - 1].x),
                       PO[k - 1].y - q.tdtsdx * (CU[k].y - CUw[k].y) - q.tdtsdy * (CV[k].y - CV[k - 1].y));
    }
}

// The same for the SW-offset staggering (shallow_tile_sw's expression trees, operand for operand): cu, cv, z look
// west / south, h looks east / north.
template <int NR>
__device__ __forceinline__ void sw_step_rows_sw(const dlesm_sw_params &q, const W2 (&U)[NR + 2], const W2 (&V)[NR + 2], const W2 (&P)[NR + 2],
                                                const W2 (&UO)[NR], const W2 (&VO)[NR], const W2 (&PO)[NR], W2 (&UN)[NR], W2 (&VN)[NR],
                                                W2 (&PN)[NR])
{
    W2 Pw[NR + 2], Vw[NR + 2], Ue[NR + 1];
#pragma unroll
    for (int k = 0; k < NR + 2; k++) Pw[k] = west(P[k]);
#pragma unroll
    for (int k = 1; k < NR + 2; k++) Vw[k] = west(V[k]);
#pragma unroll
    for (int k = 0; k < NR + 1; k++) Ue[k] = east(U[k]);
    W2 CU[NR + 1], H[NR + 1], CV[NR + 2], Z[NR + 2];
#pragma unroll
    for (int k = 0; k < NR + 1; k++) {
        CU[k] = XW(0.5 * (P[k].x + Pw[k].x) * U[k].x, 0.5 * (P[k].y + Pw[k].y) * U[k].y);
        H[k] = XW(P[k].x + 0.25 * (Ue[k].x * Ue[k].x + U[k].x * U[k].x + V[k + 1].x * V[k + 1].x + V[k].x * V[k].x),
                  P[k].y + 0.25 * (Ue[k].y * Ue[k].y + U[k].y * U[k].y + V[k + 1].y * V[k + 1].y + V[k].y * V[k].y));
    }
#pragma unroll
    for (int k = 1; k < NR + 2; k++) {
        CV[k] = XW(0.5 * (P[k].x + P[k - 1].x) * V[k].x, 0.5 * (P[k].y + P[k - 1].y) * V[k].y);
        Z[k] = XW((q.fsdx * (V[k].x - Vw[k].x) - q.fsdy * (U[k].x - U[k - 1].x)) / (Pw[k - 1].x + P[k - 1].x + P[k].x + Pw[k].x),
                  (q.fsdx * (V[k].y - Vw[k].y) - q.fsdy * (U[k].y - U[k - 1].y)) / (Pw[k - 1].y + P[k - 1].y + P[k].y + Pw[k].y));
    }
    W2 CUe[NR + 1], Ze[NR + 1], CVw[NR + 2], Hw[NR + 1];
#pragma unroll
    for (int k = 0; k < NR + 1; k++) CUe[k] = east(CU[k]);
#pragma unroll
    for (int k = 1; k < NR + 1; k++) Ze[k] = east(Z[k]);
#pragma unroll
    for (int k = 1; k < NR + 2; k++) CVw[k] = west(CV[k]);
#pragma unroll
    for (int k = 1; k < NR + 1; k++) Hw[k] = west(H[k]);
#pragma unroll
    for (int k = 1; k <= NR; k++) {
        UN[k - 1] = XW(UO[k - 1].x + q.tdts8 * (Z[k + 1].x + Z[k].x) * (CV[k + 1].x + CVw[k + 1].x + CVw[k].x + CV[k].x) -
                           q.tdtsdx * (H[k].x - Hw[k].x),
                       UO[k - 1].y + q.tdts8 * (Z[k + 1].y + Z[k].y) * (CV[k + 1].y + CVw[k + 1].y + CVw[k].y + CV[k].y) -
                           q.tdtsdx * (H[k].y - Hw[k].y));
        VN[k - 1] = XW(VO[k - 1].x - q.tdts8 * (Ze[k].x + Z[k].x) * (CUe[k].x + CU[k].x + CU[k - 1].x + CUe[k - 1].x) -
                           q.tdtsdy * (H[k].x - H[k - 1].x),
                       VO[k - 1].y - q.tdts8 * (Ze[k].y + Z[k].y) * (CUe[k].y + CU[k].y + CU[k - 1].y + CUe[k - 1].y) -
                           q.tdtsdy * (H[k].y - H[k - 1].y));
        PN[k - 1] = XW(PO[k - 1].x - q.tdtsdx * (CUe[k].x - CU[k].x) - q.tdtsdy * (CV[k + 1].x - CV[k].x),
                       PO[k - 1].y - q.tdtsdx * (CUe[k].y - CU[k].y) - q.tdtsdy * (CV[k + 1].y - CV[k].y));
    }
}

struct X2Arrays {
    const double *u, *v, *p;          // level n   (3 x 3 footprint, two cells deep)
    const double *uo, *vo, *po;       // level n-1 (read at the cell, one cell around the tile)
    double *u1, *v1, *p1;             // level n+1 (written on the box; its ring is READ by the tiles on the edge of the box)
    double *u2, *v2, *p2;             // level n+2 (written on the box)
};

// NTM bit 0: level n-1 loaded non-temporally; bit 1: both new levels stored non-temporally; bit 2: every row load of the tile is
// issued before the first use (a scheduling barrier behind the load block: without it the compiler, short of registers,
// interleaves loads and arithmetic and a wave goes through ten dependent memory round trips with five loads in flight)
// SM: the Asselin filter of the GOcean leapfrog (time_smooth) after EACH of the two steps, as dlesm_shallow_step_smooth_f64
// applies it after one: the old level the second stage uses is the filtered level n, and what is stored besides level n+2 is the
// FILTERED level n+1 (into u1, v1, p1) -- the unfiltered one never leaves the registers.  The filter is cell-local, so it
// costs no halo.  The ring of level n+1 is then taken from level n (u, v, p): a non-periodic model keeps the same fixed
// boundary values at every time level, and no array holds an unfiltered level n+1 to read it from.
// WPE: the occupancy (waves per SIMD) the register allocator is held to -- __launch_bounds__(256) alone only promises one.
// HL: lanes per side that only feed their neighbours.  One is all two steps need; with four the 112 output columns of a tile are
// seven whole 128-byte lines of the six arrays written (dlesm_shallow.hip) -- that pays for the four-row plain form (1.253 ->
// 1.208 ms at 8192^2) and costs the two-row forms 3-4 % (twice the tiles, twice the extra lanes loaded), which keep one.
template <int R, int NTM, bool SM = false, int WPE = 1, int HL = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void shallow_tile_x2(dlesm_sw_params q, int ld, int ny, int x0, int x1, int y0, int y1, int cb, int nxw,
                                                       int stack, X2Arrays a, double alpha)
{
    constexpr int X2_HALO_LANES = HL, X2_OUT_LANES = 64 - 2 * HL;
    const int lane = threadIdx.x & 63;
    int xw, strip;
    if (stack) {      // the waves of a workgroup are VERTICALLY adjacent tiles: the rows two of them share are requested by one CU
        xw = blockIdx.x % nxw;
        strip = (blockIdx.x / nxw) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    } else {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        xw = w % nxw, strip = w / nxw;
    }
    const int jb = y0 + strip * R;
    if (jb > y1) return;
    const int je = jb + R - 1 < y1 ? jb + R - 1 : y1;
    const int c = cb + xw * X2_OUT_LANES - X2_HALO_LANES + lane;             // this lane's chunk (2 columns)
    if (c - lane + X2_HALO_LANES > x1 / 2) return;                 // idle padding tile
    const int c_ld = ld / 2 - 1;
    const int cl = c < 0 ? 0 : (c > c_ld ? c_ld : c);  // halo / trailing lanes: any valid chunk
    const bool out_lane = lane >= X2_HALO_LANES && lane < 64 - X2_HALO_LANES && c <= c_ld;
    const bool m0 = out_lane && 2 * c >= x0 && 2 * c <= x1;
    const bool m1 = out_lane && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;
    const size_t col = (size_t)cl * 2;
    auto rowclamp = [&](int jj) { return jj < 0 ? 0 : (jj > ny - 1 ? ny - 1 : jj); };   // (rows off the array: loaded, never used)

    // level n: rows jb-2 .. jb+R+1 (index k); level n-1: rows jb-1 .. jb+R (index k)
    W2 U[R + 4], V[R + 4], P[R + 4], UO[R + 2], VO[R + 2], PO[R + 2];
#pragma unroll
    for (int k = 0; k < R + 4; k++) {
        const size_t o = (size_t)rowclamp(jb - 2 + k) * ld + col;
        U[k] = xld<false>(a.u + o);
        V[k] = xld<false>(a.v + o);
        P[k] = xld<false>(a.p + o);
    }
#pragma unroll
    for (int k = 0; k < R + 2; k++) {
        const size_t o = (size_t)rowclamp(jb - 1 + k) * ld + col;
        UO[k] = xld<(NTM & 1) != 0>(a.uo + o);
        VO[k] = xld<(NTM & 1) != 0>(a.vo + o);
        PO[k] = xld<(NTM & 1) != 0>(a.po + o);
    }
    if constexpr ((NTM & 4) != 0) __builtin_amdgcn_sched_barrier(0);
    // stage 1: level n+1 on rows jb-1 .. jb+R (index k), every lane's two columns
    W2 U1[R + 2], V1[R + 2], P1[R + 2];
    sw_step_rows<R + 2>(q, U, V, P, UO, VO, PO, U1, V1, P1);
    // cells of level n+1 outside the box are the fixed ring the level-n+1 arrays hold: only tiles on the edge of the box
    // (wave-uniform test) load them
    const int c0 = c - lane;                            // lane 0's chunk
    const bool rim = jb - 1 < y0 || jb + R > y1 || 2 * c0 < x0 || 2 * c0 + 127 > x1;
    if (rim) {
#pragma unroll
        for (int k = 0; k < R + 2; k++) {
            const int jj = jb - 1 + k;
            const size_t o = (size_t)rowclamp(jj) * ld + col;
            W2 ru, rv, rp;
            if constexpr (SM) { ru = U[k + 1]; rv = V[k + 1]; rp = P[k + 1]; (void)o; }      // (level n's ring, already in registers)
            else { ru = xld<false>(a.u1 + o); rv = xld<false>(a.v1 + o); rp = xld<false>(a.p1 + o); }
            const bool rin = jj >= y0 && jj <= y1;
            const bool i0 = rin && 2 * c >= x0 && 2 * c <= x1, i1 = rin && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;
            U1[k] = XW(i0 ? U1[k].x : ru.x, i1 ? U1[k].y : ru.y);
            V1[k] = XW(i0 ? V1[k].x : rv.x, i1 ? V1[k].y : rv.y);
            P1[k] = XW(i0 ? P1[k].x : rp.x, i1 ? P1[k].y : rp.y);
        }
    }
    // stage 2: level n+2 on rows jb .. jb+R-1 from level n+1 (rows jb-1 .. jb+R) and level n (rows jb .. jb+R-1) as the old one
    W2 U2[R], V2[R], P2[R], UC[R], VC[R], PC[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        UC[k] = U[k + 2]; VC[k] = V[k + 2]; PC[k] = P[k + 2];
        if constexpr (SM) {   // time_smooth after step 1: field_old = field + alpha*(field_new - 2*field + field_old), the tile's own cells
            UC[k] = XW(U[k + 2].x + alpha * (U1[k + 1].x - 2.0 * U[k + 2].x + UO[k + 1].x), U[k + 2].y + alpha * (U1[k + 1].y - 2.0 * U[k + 2].y + UO[k + 1].y));
            VC[k] = XW(V[k + 2].x + alpha * (V1[k + 1].x - 2.0 * V[k + 2].x + VO[k + 1].x), V[k + 2].y + alpha * (V1[k + 1].y - 2.0 * V[k + 2].y + VO[k + 1].y));
            PC[k] = XW(P[k + 2].x + alpha * (P1[k + 1].x - 2.0 * P[k + 2].x + PO[k + 1].x), P[k + 2].y + alpha * (P1[k + 1].y - 2.0 * P[k + 2].y + PO[k + 1].y));
        }
    }
    sw_step_rows<R>(q, U1, V1, P1, UC, VC, PC, U2, V2, P2);
    if constexpr (SM) {       // time_smooth after step 2: what is stored as the (filtered) level n+1
#pragma unroll
        for (int k = 0; k < R; k++) {
            const W2 fu = XW(U1[k + 1].x + alpha * (U2[k].x - 2.0 * U1[k + 1].x + UC[k].x), U1[k + 1].y + alpha * (U2[k].y - 2.0 * U1[k + 1].y + UC[k].y));
            const W2 fv = XW(V1[k + 1].x + alpha * (V2[k].x - 2.0 * V1[k + 1].x + VC[k].x), V1[k + 1].y + alpha * (V2[k].y - 2.0 * V1[k + 1].y + VC[k].y));
            const W2 fp = XW(P1[k + 1].x + alpha * (P2[k].x - 2.0 * P1[k + 1].x + PC[k].x), P1[k + 1].y + alpha * (P2[k].y - 2.0 * P1[k + 1].y + PC[k].y));
            U1[k + 1] = fu; V1[k + 1] = fv; P1[k + 1] = fp;
        }
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int jj = jb + k;
        if (jj > je) break;
        const size_t o = (size_t)jj * ld + (size_t)c * 2;
        if (m0 && m1) {
            xst<(NTM & 2) != 0>(a.u1 + o, U1[k + 1]);
            xst<(NTM & 2) != 0>(a.v1 + o, V1[k + 1]);
            xst<(NTM & 2) != 0>(a.p1 + o, P1[k + 1]);
            xst<(NTM & 2) != 0>(a.u2 + o, U2[k]);
            xst<(NTM & 2) != 0>(a.v2 + o, V2[k]);
            xst<(NTM & 2) != 0>(a.p2 + o, P2[k]);
        } else {
            if (m0) { a.u1[o] = U1[k + 1].x; a.v1[o] = V1[k + 1].x; a.p1[o] = P1[k + 1].x; a.u2[o] = U2[k].x; a.v2[o] = V2[k].x; a.p2[o] = P2[k].x; }
            if (m1) { a.u1[o + 1] = U1[k + 1].y; a.v1[o + 1] = V1[k + 1].y; a.p1[o + 1] = P1[k + 1].y;
                      a.u2[o + 1] = U2[k].y; a.v2[o + 1] = V2[k].y; a.p2[o + 1] = P2[k].y; }
        }
    }
}

// The SW-offset, DOUBLY PERIODIC model (the GOcean `shallow` benchmark's configuration), two steps per launch.  Nothing is a
// fixed ring here: level n+1 one cell outside the box is the periodic image of level n+1 inside, i.e. the first stage's
// result on level-n operands that lie TWO cells outside the box -- one more than the halos hold.  So a chunk (or a row) that
// lies wholly beyond the halo is loaded from where it is the image of: column i < x0 - 1 from i + Lx, i > x1 + 1 from i - Lx,
// rows likewise (16-byte loads at whatever 8-byte alignment the shift leaves); the halo cells themselves are read where they
// are.  The first stage then computes the images with the operands the opposite edge's tile uses: the same bits.  Both levels
// that come out are stored with their periodic images (shallow_tile_sw's rule: x pair first, then the y pair over the widened
// columns, field_mod.f90:1394-1464).  Precondition beyond the single step's: level n-1 carries valid halos too (the first stage
// needs it one cell around the tile); every step entry of this library that writes a level writes its images.
template <int R, int NTM, bool SM, int WPE = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void shallow_tile_sw_x2(dlesm_sw_params q, int ld, int ny, int x0, int x1, int y0, int y1, int cb, int nxw,
                                                          X2Arrays a, double alpha)
{
    constexpr int X2_HALO_LANES = 1, X2_OUT_LANES = 62;
    const int lane = threadIdx.x & 63;
    const int xw = blockIdx.x % nxw;                     // (four vertically adjacent tiles per workgroup, as shallow_tile_x2)
    const int strip = (blockIdx.x / nxw) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int jb = y0 + strip * R;
    if (jb > y1) return;
    const int je = jb + R - 1 < y1 ? jb + R - 1 : y1;
    const int c = cb + xw * X2_OUT_LANES - X2_HALO_LANES + lane;
    if (c - lane + X2_HALO_LANES > x1 / 2) return;
    const int Lx = x1 - x0 + 1, Ly = y1 - y0 + 1;
    const bool out_lane = lane >= X2_HALO_LANES && lane < 64 - X2_HALO_LANES;
    const bool m0 = out_lane && 2 * c >= x0 && 2 * c <= x1;
    const bool m1 = out_lane && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;
    // this lane's two columns, from where they are the periodic image of when they lie wholly beyond the halo
    long colw = 2L * c;
    if (2 * c + 1 < x0) colw += Lx;                      // (x0 - 2, x0 - 1) -> (x1 - 1, x1); the halo cell x0 - 1 holds x1's value anyway
    else if (2 * c > x1) colw -= Lx;
    if (colw < 0) colw = 0;                              // lanes further out than any output lane needs: any valid address
    if (colw > ld - 2) colw = ld - 2;
    auto roww = [&](int jj) {                            // rows beyond the halo rows likewise; anything else clamped into the array
        if (jj < y0 - 1) jj += Ly;
        else if (jj > y1 + 1) jj -= Ly;
        return jj < 0 ? 0 : (jj > ny - 1 ? ny - 1 : jj);
    };
    typedef double x2_d2a8 __attribute__((ext_vector_type(2), aligned(8)));
    auto ldw = [&](const double *f, int jj) {
        const x2_d2a8 t = *(const x2_d2a8 *)(f + (size_t)roww(jj) * ld + colw);
        return W2{t.x, t.y};
    };
    W2 U[R + 4], V[R + 4], P[R + 4], UO[R + 2], VO[R + 2], PO[R + 2];
#pragma unroll
    for (int k = 0; k < R + 4; k++) { U[k] = ldw(a.u, jb - 2 + k); V[k] = ldw(a.v, jb - 2 + k); P[k] = ldw(a.p, jb - 2 + k); }
#pragma unroll
    for (int k = 0; k < R + 2; k++) { UO[k] = ldw(a.uo, jb - 1 + k); VO[k] = ldw(a.vo, jb - 1 + k); PO[k] = ldw(a.po, jb - 1 + k); }
    if constexpr ((NTM & 4) != 0) __builtin_amdgcn_sched_barrier(0);
    W2 U1[R + 2], V1[R + 2], P1[R + 2];
    sw_step_rows_sw<R + 2>(q, U, V, P, UO, VO, PO, U1, V1, P1);
    W2 U2[R], V2[R], P2[R], UC[R], VC[R], PC[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        UC[k] = U[k + 2]; VC[k] = V[k + 2]; PC[k] = P[k + 2];
        if constexpr (SM) {
            UC[k] = XW(U[k + 2].x + alpha * (U1[k + 1].x - 2.0 * U[k + 2].x + UO[k + 1].x), U[k + 2].y + alpha * (U1[k + 1].y - 2.0 * U[k + 2].y + UO[k + 1].y));
            VC[k] = XW(V[k + 2].x + alpha * (V1[k + 1].x - 2.0 * V[k + 2].x + VO[k + 1].x), V[k + 2].y + alpha * (V1[k + 1].y - 2.0 * V[k + 2].y + VO[k + 1].y));
            PC[k] = XW(P[k + 2].x + alpha * (P1[k + 1].x - 2.0 * P[k + 2].x + PO[k + 1].x), P[k + 2].y + alpha * (P1[k + 1].y - 2.0 * P[k + 2].y + PO[k + 1].y));
        }
    }
    // (the filtered level n is needed ONE CELL AROUND the tile as the second stage's old level?  No: a step reads its old level
    //  at the cell itself only, and the second stage runs on the tile's own cells.)
    sw_step_rows_sw<R>(q, U1, V1, P1, UC, VC, PC, U2, V2, P2);
    if constexpr (SM) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const W2 fu = XW(U1[k + 1].x + alpha * (U2[k].x - 2.0 * U1[k + 1].x + UC[k].x), U1[k + 1].y + alpha * (U2[k].y - 2.0 * U1[k + 1].y + UC[k].y));
            const W2 fv = XW(V1[k + 1].x + alpha * (V2[k].x - 2.0 * V1[k + 1].x + VC[k].x), V1[k + 1].y + alpha * (V2[k].y - 2.0 * V1[k + 1].y + VC[k].y));
            const W2 fp = XW(P1[k + 1].x + alpha * (P2[k].x - 2.0 * P1[k + 1].x + PC[k].x), P1[k + 1].y + alpha * (P2[k].y - 2.0 * P1[k + 1].y + PC[k].y));
            U1[k + 1] = fu; V1[k + 1] = fv; P1[k + 1] = fp;
        }
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int jj = jb + k;
        if (jj > je) break;
        const int image_n = jj == y0 ? y1 + 1 : -1, image_s = jj == y1 ? y0 - 1 : -1;
        auto store3 = [&](double *fu, double *fv, double *fp, const W2 &x, const W2 &y, const W2 &z) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int jr = r == 0 ? jj : r == 1 ? image_n : image_s;
                if (jr < 0) continue;
                const size_t row = (size_t)jr * ld, o = row + (size_t)c * 2;
                if (m0 && m1) {
                    xst<(NTM & 2) != 0>(fu + o, x);
                    xst<(NTM & 2) != 0>(fv + o, y);
                    xst<(NTM & 2) != 0>(fp + o, z);
                } else {
                    if (m0) { fu[o] = x.x; fv[o] = y.x; fp[o] = z.x; }
                    if (m1) { fu[o + 1] = x.y; fv[o + 1] = y.y; fp[o + 1] = z.y; }
                }
                const int i0 = 2 * c, i1 = 2 * c + 1;      // the first / last internal column also goes to the opposite halo column
                if (m0 && i0 == x0) { fu[row + x1 + 1] = x.x; fv[row + x1 + 1] = y.x; fp[row + x1 + 1] = z.x; }
                if (m1 && i1 == x0) { fu[row + x1 + 1] = x.y; fv[row + x1 + 1] = y.y; fp[row + x1 + 1] = z.y; }
                if (m0 && i0 == x1) { fu[row + x0 - 1] = x.x; fv[row + x0 - 1] = y.x; fp[row + x0 - 1] = z.x; }
                if (m1 && i1 == x1) { fu[row + x0 - 1] = x.y; fv[row + x0 - 1] = y.y; fp[row + x0 - 1] = z.y; }
            }
        };
        store3(a.u1, a.v1, a.p1, U1[k + 1], V1[k + 1], P1[k + 1]);
        store3(a.u2, a.v2, a.p2, U2[k], V2[k], P2[k]);
    }
}

} // namespace

} // namespace dlesm

namespace dlesm {
namespace {
// the wave-tile launch of both entries below; alpha != nullptr: the filtered form
static void launch_x2(const dlesm_sw_params &q, int ld, int ny, int xstart, int xstop, int ystart, int ystop, X2Arrays a, const double *alpha,
                      hipStream_t s)
{
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    const int cb = (x0 / 2) & ~7;                        // tiles anchored on a 128-byte line of the row, as shallow_tile
    // The shape measured best at 8192^2 (scripts/shallow_x2_probe.py, profiles/r04_shallow_x2.txt): four-row tiles; the four
    // waves of a workgroup are four VERTICALLY adjacent tiles, so that of the (R+4)/R = 2 x re-read of level n only the two
    // rows above and below a 16-row band come from another workgroup -- there is no tile-to-XCD mapping to get wrong (the
    // same tiles dealt row-major under the Jacobi sweep's shape rule: 1.53 against 1.22 ms at 8192^2, within 1-7 % at
    // 2048^2 .. 6144^2 and 12288^2, profiles/r04_shallow_x2.txt); new levels stored
    // non-temporally, level n-1 loaded with the default policy (its halo rows ARE re-read; non-temporal: 1.47 ms).
    // sw_x2_rows / _nt / _stack / _pad select the comparison forms (lab build).
    // The filtered form, THREE-row tiles held to two waves per SIMD (256 registers, no scratch): 1.26-1.27 ms against 1.31-1.33
    // for two-row tiles with every load issued first (210 registers) on the same box -- fewer rows of level n requested per row
    // written ((R+4)/R: 2.33 against 3).  Four-row tiles do not fit: capped 164-336 B of scratch, uncapped one wave per SIMD,
    // also with the rows of level n-1 the filter needs loaded a second time instead of kept live.
    int R = alpha ? 3 : 4, nt = 2, stack = 4, pad = 0;
    if (kLab) {
        const int rows = tuning("sw_x2_rows", 0);        // (0: the product's height for the form)
        if (rows == 2 || rows == 4 || rows == 6 || (rows == 3 && alpha)) R = rows;
        nt = tuning("sw_x2_nt", R == 2 && alpha ? 6 : nt) & 7;
        stack = tuning("sw_x2_stack", 4);
        pad = tuning("sw_x2_pad", 0);
    }
    // the four-row plain form on whole-line tiles (HL = 4) when every row of every array starts on a 128-byte line
    bool lines = ld % 16 == 0;
    for (const double *f : {a.u, a.v, a.p, a.uo, a.vo, a.po, (const double *)a.u1, (const double *)a.v1, (const double *)a.p1,
                            (const double *)a.u2, (const double *)a.v2, (const double *)a.p2})
        lines = lines && ((uintptr_t)f % 128 == 0);
    const int out_lanes = (!alpha && R == 4 && lines) ? 56 : 62;
    int nxw = (x1 / 2 - cb + out_lanes) / out_lanes, tpb = 4;
    if (stack) {
        nxw += pad;
        tpb = stack == 2 ? 2 : 4;
    } else {
        choose_block_shape(&nxw, &tpb);
        if (tpb > 4) tpb = 4;
    }
    const int strips = (y1 - y0 + R) / R;
    const unsigned grid = stack ? (unsigned)((long)nxw * ((strips + tpb - 1) / tpb)) : (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
    const double al = alpha ? *alpha : 0.0;
#define DLESM_X2(RR, NN)                                                                                                                                  \
    do {                                                                                                                                                  \
        if (alpha) hipLaunchKernelGGL((shallow_tile_x2<RR, NN, true>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al); \
        else if (RR == 4 && lines) hipLaunchKernelGGL((shallow_tile_x2<RR, NN, false, 1, (RR == 4 ? 4 : 1)>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al); \
        else hipLaunchKernelGGL((shallow_tile_x2<RR, NN, false, 1, 1>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al); \
    } while (0)
#ifdef DLESM_LAB
#define DLESM_X2R(RR)                                                                                                  \
    switch (nt) {                                                                                                      \
    case 0: DLESM_X2(RR, 0); break; case 3: DLESM_X2(RR, 3); break; case 4: DLESM_X2(RR, 4); break;                   \
    case 6: DLESM_X2(RR, 6); break; case 7: DLESM_X2(RR, 7); break; default: DLESM_X2(RR, 2); break;                  \
    }
    if (R == 3) hipLaunchKernelGGL((shallow_tile_x2<3, 2, true, 2>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al);
    else if (R == 2) { DLESM_X2R(2) } else if (R == 6) { DLESM_X2R(6) } else { DLESM_X2R(4) }
#undef DLESM_X2R
#else
    (void)nt;
    if (alpha) hipLaunchKernelGGL((shallow_tile_x2<3, 2, true, 2>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al);
    else if (lines) hipLaunchKernelGGL((shallow_tile_x2<4, 2, false, 1, 4>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al);
    else hipLaunchKernelGGL((shallow_tile_x2<4, 2, false, 1, 1>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, stack, a, al);
#endif
#undef DLESM_X2
}
// the SW-offset doubly periodic form (shallow_tile_sw_x2): four vertically adjacent tiles per workgroup; tile heights and load
// order as launch_x2 chose them for the NE forms
static void launch_sw_x2(const dlesm_sw_params &q, int ld, int ny, int xstart, int xstop, int ystart, int ystop, X2Arrays a, const double *alpha,
                         hipStream_t s)
{
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    const int cb = (x0 / 2) & ~7;
    const int nxw = (x1 / 2 - cb + 62) / 62, tpb = 4;
    // plain form: two-row tiles held to two waves per SIMD.  Left to itself (a 256-thread launch bound only promises one wave per
    // SIMD) the allocator takes 260 registers for two-row and 346 for four-row tiles -- one wave per SIMD, 1.555 ms per launch at
    // 8192^2; capped at 256: 1.357 ms (four-row tiles capped: 320 B of scratch, 2.06 ms).  sw_x2_sw_form = 1 (lab): the uncapped
    // four-row form.
    // filtered form: three-row tiles at two waves per SIMD (248 registers), 1.30-1.33 ms against 1.38-1.39 for two-row tiles
    // (sw_x2_sw_form = 1 in the lab build: the two-row form)
    const int form = kLab ? tuning("sw_x2_sw_form", 0) : 0;
    const int R = alpha ? (form == 1 ? 2 : 3) : (form == 1 ? 4 : 2);
    const int strips = (y1 - y0 + R) / R;
    const unsigned grid = (unsigned)((long)nxw * ((strips + tpb - 1) / tpb));
#ifdef DLESM_LAB
    if (alpha && form == 1) hipLaunchKernelGGL((shallow_tile_sw_x2<2, 6, true>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, a, *alpha);
    else
#endif
    if (alpha) hipLaunchKernelGGL((shallow_tile_sw_x2<3, 2, true, 2>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, a, *alpha);
#ifdef DLESM_LAB
    else if (form == 1) hipLaunchKernelGGL((shallow_tile_sw_x2<4, 2, false>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, a, 0.0);
#endif
    else hipLaunchKernelGGL((shallow_tile_sw_x2<2, 2, false, 2>), dim3(grid), dim3(64 * tpb), 0, s, q, ld, ny, x0, x1, y0, y1, cb, nxw, a, 0.0);
}
} // namespace
} // namespace dlesm

using namespace dlesm;

// the twelve arrays of a two-step call are ld x ny doubles each and must not overlap (a shifted alias would be read and written
// by different tiles of one launch)
static int x2_disjoint(const char *who, const double *const (&all)[12], int ld, int ny)
{
    const uintptr_t bytes = (uintptr_t)ld * (uintptr_t)ny * sizeof(double);
    for (int i = 0; i < 12; i++)
        for (int j = i + 1; j < 12; j++) {
            const uintptr_t a = (uintptr_t)all[i], b = (uintptr_t)all[j];
            DLESM_REQUIRE((a < b ? b - a : a - b) >= bytes, "%s: the twelve arrays must not overlap (arguments %d and %d do)", who, i, j);
        }
    return DLESM_OK;
}

// Two leapfrog steps, one launch (NE offset, fixed boundary ring): level n+1 into (unew, vnew, pnew), level n+2 into
// (unew2, vnew2, pnew2) -- bit for bit what
//     dlesm_shallow_step_f64(q, ..., u, v, p, uold, vold, pold, unew, vnew, pnew);
//     dlesm_shallow_step_f64(q, ..., unew, vnew, pnew, u, v, p, unew2, vnew2, pnew2);
// leave behind, INCLUDING what the second call reads outside the box: the ring of unew, vnew, pnew (which no step writes) must
// hold the boundary values before the call, as for the two calls.  Arrays that miss the conditions of the wave-tile kernel
// (16-byte aligned bases; an even leading dimension or a box that ends inside the last whole 2-column chunk) take those two calls.
extern "C" int dlesm_shallow_step_x2_f64(const dlesm_sw_params *q, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                                         const double *u, const double *v, const double *p, const double *uold, const double *vold,
                                         const double *pold, double *unew, double *vnew, double *pnew, double *unew2, double *vnew2,
                                         double *pnew2, void *stream)
{
    clear_error();
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(q && u && v && p && uold && vold && pold && unew && vnew && pnew && unew2 && vnew2 && pnew2, "null pointer");
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("dlesm_shallow_step_x2_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    const double *all[12] = {u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2};
    if (int rc = x2_disjoint("dlesm_shallow_step_x2_f64", all, ld, ny)) return rc;
    bool aligned = ld % 2 == 0 || (xstop - 1) + 1 <= 2 * (ld / 2) - 1;
    for (const double *f : all) aligned = aligned && ((uintptr_t)f % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (!aligned || tuning("sw_kernel", 0) != 0 || !tuning("sw_x2_fused", 1)) {      // the definition: two single steps
        if (int rc = dlesm_shallow_step_f64(q, ld, ny, xstart, xstop, ystart, ystop, u, v, p, uold, vold, pold, unew, vnew, pnew, stream)) return rc;
        return dlesm_shallow_step_f64(q, ld, ny, xstart, xstop, ystart, ystop, unew, vnew, pnew, u, v, p, unew2, vnew2, pnew2, stream);
    }
    launch_x2(*q, ld, ny, xstart, xstop, ystart, ystop, X2Arrays{u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2}, nullptr, s);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// Two FILTERED leapfrog steps, one launch: two whole time steps of the GOcean loop (update + time_smooth of the old level, twice).
// With levels n (u, v, p) and n-1 (uold, vold, pold: the filtered one, as the loop keeps it) it leaves level n+2 in
// (unew2, vnew2, pnew2) and the FILTERED level n+1 in (uold2, vold2, pold2) -- what two calls of dlesm_shallow_step_smooth_f64
// with the usual rotation leave as the new current and old levels -- at 6 arrays read + 6 written per TWO steps: 48 B/cell/step
// against 96 for the one-launch filtered step and 168 for step + three filter launches.  The inputs are not modified; the ring of
// the unfiltered level n+1 the second stage reads outside the box is taken from u, v, p (a fixed boundary ring is the same at
// every time level).  Time loop: ping-pong (cur, old) <-> (unew2.., uold2..).  Arrays that miss the wave-tile conditions take the
// definition through three stream-ordered scratch arrays.
extern "C" int dlesm_shallow_step_smooth_x2_f64(const dlesm_sw_params *q, double alpha, int ld, int ny, int xstart, int xstop, int ystart,
                                                int ystop, const double *u, const double *v, const double *p, const double *uold,
                                                const double *vold, const double *pold, double *unew2, double *vnew2, double *pnew2,
                                                double *uold2, double *vold2, double *pold2, void *stream)
{
    clear_error();
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(q && u && v && p && uold && vold && pold && unew2 && vnew2 && pnew2 && uold2 && vold2 && pold2, "null pointer");
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("dlesm_shallow_step_smooth_x2_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    const double *all[12] = {u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2};
    if (int rc = x2_disjoint("dlesm_shallow_step_smooth_x2_f64", all, ld, ny)) return rc;
    bool aligned = ld % 2 == 0 || (xstop - 1) + 1 <= 2 * (ld / 2) - 1;
    for (const double *f : all) aligned = aligned && ((uintptr_t)f % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (aligned && tuning("sw_kernel", 0) == 0 && tuning("sw_x2_fused", 1)) {
        // (X2Arrays: u1.. = where the stored level n+1 goes -- here the filtered one; u2.. = level n+2)
        launch_x2(*q, ld, ny, xstart, xstop, ystart, ystop, X2Arrays{u, v, p, uold, vold, pold, uold2, vold2, pold2, unew2, vnew2, pnew2}, &alpha, s);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    // the definition: old2 <- old; T <- ring of u (the level-n+1 ring); step_smooth(u, old2 -> filtered n, T = n+1);
    // step_smooth(T, old2 -> filtered n+1, new2 = n+2)
    const size_t bytes = (size_t)ld * ny * sizeof(double);
    double *t[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++)
        if (hipMallocAsync((void **)&t[k], bytes, s) != hipSuccess) {
            for (int q2 = 0; q2 < k; q2++) (void)hipFreeAsync(t[q2], s);
            return fail(DLESM_EHIP, "dlesm_shallow_step_smooth_x2_f64: no scratch memory for the intermediate time level");
        }
    const double *src[6] = {uold, vold, pold, u, v, p};
    double *dst[6] = {uold2, vold2, pold2, t[0], t[1], t[2]};
    int rc = DLESM_OK;
    for (int k = 0; k < 6 && !rc; k++)
        if (hipMemcpyAsync(dst[k], src[k], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = fail(DLESM_EHIP, "dlesm_shallow_step_smooth_x2_f64: copy failed");
    if (!rc) rc = dlesm_shallow_step_smooth_f64(q, alpha, ld, ny, xstart, xstop, ystart, ystop, u, v, p, uold2, vold2, pold2, t[0], t[1], t[2], stream);
    if (!rc) rc = dlesm_shallow_step_smooth_f64(q, alpha, ld, ny, xstart, xstop, ystart, ystop, t[0], t[1], t[2], uold2, vold2, pold2, unew2, vnew2, pnew2, stream);
    for (int k = 0; k < 3; k++) (void)hipFreeAsync(t[k], s);
    return rc;
}

// ---- the SW-offset, doubly periodic model (the GOcean `shallow` benchmark's configuration) --------------------------------------
static int sw_x2_common(const char *who, const dlesm_region *internal, int ld, int ny, const double *const (&all)[12], bool *aligned)
{
    DLESM_REQUIRE(internal != nullptr, "null pointer");
    for (const double *f : all) DLESM_REQUIRE(f != nullptr, "null pointer");
    if (int rc = check_box(who, ld, ny, internal->xstart, internal->xstop, internal->ystart, internal->ystop, 1)) return rc;
    if (int rc = x2_disjoint(who, all, ld, ny)) return rc;
    *aligned = ld % 2 == 0 || (internal->xstop - 1) + 1 <= 2 * (ld / 2) - 1;
    for (const double *f : all) *aligned = *aligned && ((uintptr_t)f % 16 == 0);
    return DLESM_OK;
}

// Two steps of the SW-offset periodic model per launch: == two calls of dlesm_shallow_step_sw_periodic_f64 (level n+1 with its
// periodic images into unew.., level n+2 with its images into unew2..).  Beyond what those calls ask for, level n-1 must carry
// valid periodic halos too (every step entry of this library that writes a level writes its images).  Anything but a doubly
// periodic box, or arrays that miss the wave-tile conditions, takes the two calls.
extern "C" int dlesm_shallow_step_sw_x2_periodic_f64(const dlesm_sw_params *q, int ld, int ny, const dlesm_region *internal, int bc_x,
                                                     int bc_y, const double *u, const double *v, const double *p, const double *uold,
                                                     const double *vold, const double *pold, double *unew, double *vnew, double *pnew,
                                                     double *unew2, double *vnew2, double *pnew2, void *stream)
{
    clear_error();
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(q != nullptr, "null pointer");
    const double *all[12] = {u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2};
    bool aligned = false;
    if (int rc = sw_x2_common("dlesm_shallow_step_sw_x2_periodic_f64", internal, ld, ny, all, &aligned)) return rc;
    if (internal->xstop < internal->xstart || internal->ystop < internal->ystart) return DLESM_OK;
    const bool both = bc_x == DLESM_BC_PERIODIC && bc_y == DLESM_BC_PERIODIC;
    // (a box narrower or lower than two cells has images that wrap more than once: the two calls)
    if (aligned && both && internal->nx >= 2 && internal->ny >= 2 && tuning("sw_kernel", 0) == 0 && tuning("sw_x2_fused", 1)) {
        launch_sw_x2(*q, ld, ny, internal->xstart, internal->xstop, internal->ystart, internal->ystop,
                     X2Arrays{u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2}, nullptr, (hipStream_t)stream);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    if (int rc = dlesm_shallow_step_sw_periodic_f64(q, ld, ny, internal, bc_x, bc_y, u, v, p, uold, vold, pold, unew, vnew, pnew, stream)) return rc;
    return dlesm_shallow_step_sw_periodic_f64(q, ld, ny, internal, bc_x, bc_y, unew, vnew, pnew, u, v, p, unew2, vnew2, pnew2, stream);
}

// Two WHOLE time steps of the GOcean `shallow` benchmark per launch: update + time_smooth of the old level + periodic images,
// twice.  In: level n and the filtered level n-1, with valid periodic halos, untouched.  Out: level n+2 (unew2..) and the filtered
// level n+1 (uold2..), each with its periodic images: what two calls of dlesm_shallow_step_sw_smooth_periodic_f64 with the loop's
// rotation leave as the new current and old levels, bit for bit, at 48 instead of 96 B/cell/step.  Ping-pong between the two sextets.
extern "C" int dlesm_shallow_step_sw_smooth_x2_periodic_f64(const dlesm_sw_params *q, double alpha, int ld, int ny,
                                                            const dlesm_region *internal, int bc_x, int bc_y, const double *u,
                                                            const double *v, const double *p, const double *uold, const double *vold,
                                                            const double *pold, double *unew2, double *vnew2, double *pnew2,
                                                            double *uold2, double *vold2, double *pold2, void *stream)
{
    clear_error();
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(q != nullptr, "null pointer");
    const double *all[12] = {u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2};
    bool aligned = false;
    if (int rc = sw_x2_common("dlesm_shallow_step_sw_smooth_x2_periodic_f64", internal, ld, ny, all, &aligned)) return rc;
    if (internal->xstop < internal->xstart || internal->ystop < internal->ystart) return DLESM_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool both = bc_x == DLESM_BC_PERIODIC && bc_y == DLESM_BC_PERIODIC;
    if (aligned && both && internal->nx >= 2 && internal->ny >= 2 && tuning("sw_kernel", 0) == 0 && tuning("sw_x2_fused", 1)) {
        launch_sw_x2(*q, ld, ny, internal->xstart, internal->xstop, internal->ystart, internal->ystop,
                     X2Arrays{u, v, p, uold, vold, pold, uold2, vold2, pold2, unew2, vnew2, pnew2}, &alpha, s);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    // the definition, through three stream-ordered scratch arrays for the unfiltered level n+1
    const size_t bytes = (size_t)ld * ny * sizeof(double);
    double *t[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++)
        if (hipMallocAsync((void **)&t[k], bytes, s) != hipSuccess) {
            for (int q2 = 0; q2 < k; q2++) (void)hipFreeAsync(t[q2], s);
            return fail(DLESM_EHIP, "dlesm_shallow_step_sw_smooth_x2_periodic_f64: no scratch memory for the intermediate time level");
        }
    const double *src[3] = {uold, vold, pold};
    double *dst[3] = {uold2, vold2, pold2};
    int rc = DLESM_OK;
    for (int k = 0; k < 3 && !rc; k++)
        if (hipMemcpyAsync(dst[k], src[k], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = fail(DLESM_EHIP, "dlesm_shallow_step_sw_smooth_x2_periodic_f64: copy failed");
    if (!rc) rc = dlesm_shallow_step_sw_smooth_periodic_f64(q, alpha, ld, ny, internal, bc_x, bc_y, u, v, p, uold2, vold2, pold2, t[0], t[1], t[2], stream);
    if (!rc) rc = dlesm_shallow_step_sw_smooth_periodic_f64(q, alpha, ld, ny, internal, bc_x, bc_y, t[0], t[1], t[2], uold2, vold2, pold2, unew2, vnew2, pnew2, stream);
    for (int k = 0; k < 3; k++) (void)hipFreeAsync(t[k], s);
    return rc;
}
