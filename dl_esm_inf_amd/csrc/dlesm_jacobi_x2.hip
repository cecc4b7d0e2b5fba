// T Jacobi-5 time steps in one sweep (temporal blocking), T = 2..8.
//
//   t_0 = in
//   t_s(i,j) = J(t_{s-1})(i,j)  for (i,j) in the stage box E_s,  t_{s-1}(i,j) elsewhere   (s = 1..T-1)
//   out(i,j) = J(t_{T-1})(i,j)  for (i,j) in the output box B
//
// with J(f)(i,j) = 0.25*((f(i-1,j)+f(i+1,j)) + (f(i,j-1)+f(i,j+1))), i.e. exactly T
// dlesm_stencil5_f64 calls through ping-pong buffers that start as copies of `in`.
// One tile: every E_s is the tile's box (the boundary ring stays fixed).  Distributed: E_s is
// the tile's box grown by T-s cells towards every neighbouring tile, `in` carrying depth-T
// halos.  The intermediates never touch memory: 16 B per cell for T steps instead of 16*T.
// Same expression tree per step as the single-step kernel, so the result is bit-identical to
// T single steps.
//
// Wave tile: 64 lanes x 2 doubles x (R+2T) input rows; stage s produces R+2(T-s) rows in place
// in registers, a column narrower on each side than the stage before, so ceil(T/2) lanes on
// each side of the wave are halo lanes and 64-2*ceil(T/2) lanes store R output rows.  West/east
// operands are whole-wave shifts by one lane (DPP on the VALU, or ds_bpermute).  Row-major
// linear sweep and block-shape rule as in jacobi5_tile.
#include <algorithm>

#include "dlesm_internal.h"

namespace dlesm {

struct XtBoxes {           // 0-based inclusive
    int x0, x1, y0, y1;     // output box B
    int ex0, ex1, ey0, ey1; // last intermediate box E_{T-1}
    int gw, ge, gs, gn;     // 0/1: E_s = E_{T-1} grown by (T-1-s) cells on that side
};

typedef double xt_d2 __attribute__((ext_vector_type(2)));

// from this many fused steps on, tile indices are computed on the scalar unit (A/B builds: 2 or 99)
#ifndef DLESM_XT_SCALAR_FROM
#define DLESM_XT_SCALAR_FROM 3
#endif

// Stages 1..T-1 in place on the register rows.  SEL: the wave touches the edge of a stage box,
// so every value is selected between J(previous stage) and the previous stage itself; waves
// wholly inside the last stage box (almost all of them) skip the selects.  `q` is 0.25, passed
// at run time so that the scaling is one full-rate v_mul_f64 (a literal becomes v_ldexp_f64).
template <int T, int R, bool DPP, bool SEL>
__device__ __forceinline__ void xt_stages(xt_d2 (&v)[R + 2 * T], const XtBoxes &b, int c, int jb, double q)
{
#pragma unroll
    for (int s = 1; s < T; s++) {
        // after this stage v[u] holds t_s on row jb-(T-s)+u
        const int g = T - 1 - s;
        const int sx0 = b.ex0 - b.gw * g, sx1 = b.ex1 + b.ge * g, sy0 = b.ey0 - b.gs * g, sy1 = b.ey1 + b.gn * g;
        const bool c0 = 2 * c >= sx0 && 2 * c <= sx1, c1 = 2 * c + 1 >= sx0 && 2 * c + 1 <= sx1;
#pragma unroll
        for (int u = 0; u < R + 2 * (T - s); u++) {
            const xt_d2 mid = v[u + 1];
            const double west = from_lower<DPP>(mid.y), east = from_upper<DPP>(mid.x);
            const double tx = q * ((west + mid.y) + (v[u].x + v[u + 2].x));
            const double ty = q * ((mid.x + east) + (v[u].y + v[u + 2].y));
            if constexpr (SEL) {
                const int jt = jb - (T - s) + u;
                const bool rowin = jt >= sy0 && jt <= sy1;
                v[u].x = (rowin && c0) ? tx : mid.x;
                v[u].y = (rowin && c1) ? ty : mid.y;
            } else {
                v[u].x = tx;
                v[u].y = ty;
            }
        }
    }
}

template <int T, int R, bool DPP>
__global__ __launch_bounds__(512) void jacobi5xt_tile(const double *__restrict__ in, double *__restrict__ out, int ld, int ny, XtBoxes b,
                    int cb, int nxw, double q)
{
    typedef xt_d2 d2;
    constexpr int H = (T + 1) / 2;                      // halo lanes per side
    const int lane = threadIdx.x & 63;
    // the wave's number inside the workgroup as a SCALAR: everything derived from it (tile
    // position, row numbers, row base addresses, the rim test) then lives in SGPRs and the row
    // loads take the "scalar base + lane offset" form instead of one 64-bit VGPR address per row
    int wv = threadIdx.x >> 6;
    if constexpr (T >= DLESM_XT_SCALAR_FROM) wv = __builtin_amdgcn_readfirstlane(wv);
    const int w = blockIdx.x * (blockDim.x >> 6) + wv;
    int xw = w % nxw, strip = w / nxw;
    if constexpr (T >= DLESM_XT_SCALAR_FROM) {
        xw = __builtin_amdgcn_readfirstlane(xw);       // (the integer division runs on the VALU)
        strip = __builtin_amdgcn_readfirstlane(strip);
    }
    const int jb = b.y0 + strip * R;
    if (jb > b.y1) return;
    int je = jb + R - 1;
    if (je > b.y1) je = b.y1;
    const int c_w = cb + xw * (64 - 2 * H) - H;          // chunk of lane 0 (wave-uniform)
    const int c = c_w + lane;
    if (c_w + H > b.x1 / 2) return;                     // idle padding tile
    const int c_ld = ld / 2 - 1;
    const int cl = c < 0 ? 0 : (c > c_ld ? c_ld : c);
    const bool ol = lane >= H && lane <= 63 - H && c <= c_ld;
    const bool m0 = ol && 2 * c >= b.x0 && 2 * c <= b.x1;
    const bool m1 = ol && 2 * c + 1 >= b.x0 && 2 * c + 1 <= b.x1;
    const unsigned lane_off = (unsigned)cl * 16u;       // bytes; a row is < 4 GiB
    d2 v[R + 2 * T];
#pragma unroll
    for (int u = 0; u < R + 2 * T; u++) {
        int jj = jb - T + u;
        if (jj > je + T) jj = je + T;                   // short last strip: loaded again, never used
        jj = jj < 0 ? 0 : (jj > ny - 1 ? ny - 1 : jj);  // rows outside the array feed discarded values only
        const char *row = (const char *)(in + (size_t)jj * ld);     // wave-uniform
        v[u] = *(const d2 *)(row + lane_off);
    }
    // every column of the wave and every intermediate row it computes inside the LAST (smallest)
    // stage box: no value is ever "carried", the selects can go
    const bool inner = 2 * c_w >= b.ex0 && 2 * (c_w + 63) + 1 <= b.ex1 && jb - (T - 1) >= b.ey0 &&
                       jb + R - 1 + (T - 1) <= b.ey1;
    if (inner) xt_stages<T, R, DPP, false>(v, b, c, jb, q);
    else xt_stages<T, R, DPP, true>(v, b, c, jb, q);
#pragma unroll
    for (int u = 0; u < R; u++) {
        const d2 mid = v[u + 1];
        const double west = from_lower<DPP>(mid.y), east = from_upper<DPP>(mid.x);
        if (jb + u <= je) {
            const double o0 = q * ((west + mid.y) + (v[u].x + v[u + 2].x));
            const double o1 = q * ((mid.x + east) + (v[u].y + v[u + 2].y));
            double *po = (double *)((char *)(out + (size_t)(jb + u) * ld) + (size_t)(unsigned)c * 16u);
            if (m0 && m1) *(d2 *)po = d2{o0, o1};
            else {
                if (m0) po[0] = o0;
                if (m1) po[1] = o1;
            }
        }
    }
}

// Two steps, one cell per thread, neighbours through L1/L2: used when the arrays do not meet the
// 16-byte lane conditions of the tile kernel.  0-based inclusive boxes.
__global__ __launch_bounds__(256) void jacobi5x2_direct(const double *__restrict__ in,
                                                        double *__restrict__ out, int ld, int x0, int x1,
                                                        int y0, int y1, int ex0, int ex1, int ey0, int ey1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x, j = y0 + blockIdx.y;
    if (i > x1 || j > y1) return;
    auto T1 = [&](int ii, int jj) -> double {
        const size_t o = (size_t)jj * ld + ii;
        if (ii < ex0 || ii > ex1 || jj < ey0 || jj > ey1) return in[o];
        return 0.25 * ((in[o - 1] + in[o + 1]) + (in[o - ld] + in[o + ld]));
    };
    out[(size_t)j * ld + i] = 0.25 * ((T1(i - 1, j) + T1(i + 1, j)) + (T1(i, j - 1) + T1(i, j + 1)));
}

#ifdef DLESM_LAB      // the pipeline form (j5xt_march = 1): a comparison point, libdlesm_hip_lab.so only
// ---------------------------------------------------------------------------------------------
// The same T steps as a PIPELINE marching in y (no vertical redundancy): a wave owns 128 columns
// and walks up a strip of rows; level s = 1..T keeps only its three newest rows in registers, and
// every new input row pushes one new row through all T levels -- level s computes row j-s from
// the rows j-s-1, j-s, j-s+1 of level s-1 -- so that row j-T of the result leaves the pipeline.
// Work per cell and step is one stencil evaluation (the tile kernel above evaluates (R+2T)/R of
// them); the price is the 2T-row run-in of every strip and long-lived waves.
//
// Only for the part of the box where nothing is ever "carried": every column of the wave and
// every row it touches lies inside the last stage box, so there are no selects and no masks; the
// rim of the box is left to jacobi5xt_tile (launch_xt_region below).  Input rows are prefetched
// P-3 rows ahead through a ring of P registers rows; all register arrays are indexed statically
// (the row loop is unrolled P times, P a multiple of 3).
// ---------------------------------------------------------------------------------------------
template <bool DPP>
__device__ __forceinline__ xt_d2 xt_row(const xt_d2 &south, const xt_d2 &mid, const xt_d2 &north, double q)
{
    const double west = from_lower<DPP>(mid.y), east = from_upper<DPP>(mid.x);
    xt_d2 r;
    r.x = q * ((west + mid.y) + (south.x + north.x));
    r.y = q * ((mid.x + east) + (south.y + north.y));
    return r;
}

template <int T, int P, bool DPP>
__global__ __launch_bounds__(256) void jacobi5xt_march(const double *__restrict__ in, double *__restrict__ out,
                                                      int ld, int cb, int ntx, int S, int Y0, int Y1, double q, int perm)
{
    typedef xt_d2 d2;
    static_assert(P % 3 == 0 && P >= 6, "ring size");
    constexpr int H = (T + 1) / 2, OL = 64 - 2 * H, D = P - 3;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = blockIdx.x * (blockDim.x >> 6) + wv;
    const int xw = __builtin_amdgcn_readfirstlane((int)(((long)(w % ntx) * perm) % ntx)), strip = __builtin_amdgcn_readfirstlane(w / ntx);
    const int Jb = Y0 + strip * S;
    if (Jb > Y1) return;
    const int Je = Jb + S - 1 < Y1 ? Jb + S - 1 : Y1;
    const unsigned lane_off = (unsigned)(cb + xw * OL + lane) * 16u;
    const bool ol = lane >= H && lane <= 63 - H;
    const int jfirst = Jb - T, jlast = Je + T;           // input rows of this strip
    const int nit = jlast - jfirst + 1;
    auto load_row = [&](int j) {
        const char *row = (const char *)(in + (size_t)(j < jlast ? j : jlast) * ld);
        return *(const d2 *)(row + lane_off);
    };
    d2 Q[P], W[T > 1 ? T : 2][3];
#pragma unroll
    for (int i = 0; i < P; i++) Q[i] = d2{0.0, 0.0};
#pragma unroll
    for (int s = 0; s < (T > 1 ? T : 2); s++)
#pragma unroll
        for (int i = 0; i < 3; i++) W[s][i] = d2{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < D; i++) Q[i] = load_row(jfirst + i);
    for (int k0 = 0; k0 < nit; k0 += P) {
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int k = k0 + i;                        // this iteration's input row is jfirst + k
            Q[(i + D) % P] = load_row(jfirst + k + D);   // overwrites row k-3, which is dead
            d2 nw = xt_row<DPP>(Q[(i + P - 2) % P], Q[(i + P - 1) % P], Q[i], q);
#pragma unroll
            for (int s = 2; s <= T; s++) {
                W[s - 1][i % 3] = nw;                    // level s-1 gets its row j-(s-1), the oldest goes
                nw = xt_row<DPP>(W[s - 1][(i + 1) % 3], W[s - 1][(i + 2) % 3], W[s - 1][i % 3], q);
            }
            const int jo = jfirst + k - T;               // the row that leaves the pipeline
            if (jo >= Jb && jo <= Je && ol)
                *(d2 *)((char *)(out + (size_t)jo * ld) + lane_off) = nw;
        }
    }
}

template <int T, bool DPP>
static void launch_xt(const double *in, double *out, int ld, int ny, const XtBoxes &b, int R, hipStream_t s);

// The box as marching interior + four rim bands of tile launches.  Returns false when the box has
// no room for the marching kernel (then nothing has been launched).
template <int T, bool DPP>
static bool launch_xt_region(const double *in, double *out, int ld, int ny, const XtBoxes &b, int R,
                             hipStream_t s)
{
    constexpr int H = (T + 1) / 2, OL = 64 - 2 * H;
    if (b.ex1 < b.ex0 || b.ey1 < b.ey0) return false;
    // rows: every level's rows of a strip inside the last stage box
    const int my0 = std::max(b.y0, b.ey0 + T - 1), my1 = std::min(b.y1, b.ey1 - (T - 1));
    // columns: lane 0 of the first tile; all 128 columns of a tile inside the last stage box, output
    // lanes inside the output box
    const int c0 = std::max((b.ex0 + 1) / 2, (b.x0 + 1) / 2 - H);
    const int c_max = std::min((b.ex1 - 1) / 2 - 63, (b.x1 - 1) / 2 - 63 + H);   // last admissible lane-0 chunk
    if (my1 - my0 + 1 < 4 * T || c_max < c0) return false;
    const int ntx = (c_max - c0) / OL + 1;
    const int mx0 = 2 * (c0 + H), mx1 = 2 * (c0 + (ntx - 1) * OL + 63 - H) + 1;   // columns the march writes
    if (ntx < 8) return false;
    // strips: one per resident wave slot, so that all waves march side by side for the whole launch
    const int slots = tuning("j5xt_march_slots", 3072);
    int nstrips = slots / ntx;
    if (nstrips < 1) nstrips = 1;
    int S = (my1 - my0 + nstrips) / nstrips;
    if (S < 8 * T) S = 8 * T;                            // run-in of 2T rows per strip: at most 25 %
    nstrips = (my1 - my0 + S) / S;
    const long waves = (long)ntx * nstrips;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    // experiment: tile number -> column position through a multiplicative permutation (1 = identity)
    int perm = tuning("j5xt_march_perm", 1);
    if (perm < 1 || std::__gcd(perm, ntx) != 1) perm = 1;
    const int P = tuning("j5xt_march_ring", 9);
    if (P == 6)
        hipLaunchKernelGGL((jacobi5xt_march<T, 6, DPP>), dim3(grid), dim3(256), 0, s, in, out, ld, c0, ntx, S, my0, my1, 0.25, perm);
    else if (P == 12)
        hipLaunchKernelGGL((jacobi5xt_march<T, 12, DPP>), dim3(grid), dim3(256), 0, s, in, out, ld, c0, ntx, S, my0, my1, 0.25, perm);
    else
        hipLaunchKernelGGL((jacobi5xt_march<T, 9, DPP>), dim3(grid), dim3(256), 0, s, in, out, ld, c0, ntx, S, my0, my1, 0.25, perm);
    // the rim: south and north bands over the full width, west and east bands beside the march
    auto band = [&](int x0, int x1, int y0, int y1) {
        if (x1 < x0 || y1 < y0) return;
        XtBoxes t = b;
        t.x0 = x0; t.x1 = x1; t.y0 = y0; t.y1 = y1;
        launch_xt<T, DPP>(in, out, ld, ny, t, R, s);
    };
    band(b.x0, b.x1, b.y0, my0 - 1);
    band(b.x0, b.x1, my1 + 1, b.y1);
    band(b.x0, mx0 - 1, my0, my1);
    band(mx1 + 1, b.x1, my0, my1);
    return true;
}

#endif // DLESM_LAB

template <int T, bool DPP>
static void launch_xt(const double *in, double *out, int ld, int ny, const XtBoxes &b, int R, hipStream_t s)
{
    constexpr int OL = 64 - 2 * ((T + 1) / 2);
    const int cb = b.x0 / 2, c_last = b.x1 / 2;
    int nxw = (c_last - cb + OL) / OL, tpb = 4;
    // measured best at 16384^2 (profiles/r01_sweep_fused.txt): 4, 6, 8, 8, 12, 16, 16 rows for
    // T = 2..8; capping the VGPRs for more waves per SIMD made no difference (2, 3, 4 waves tried)
    constexpr int best_rows[9] = {0, 8, 4, 6, 8, 8, 12, 16, 16};
#ifdef DLESM_LAB
    if (R != 2 && R != 4 && R != 6 && R != 8 && R != 12 && R != 16) R = best_rows[T];
    if (T > 4 && R < 8) R = 8;                           // deep fusions: tall tiles only
    if (T < 4 && R > 8) R = 8;
#else
    R = best_rows[T];                                    // the product holds the measured-best height of each depth only
#endif
    const int strips = (b.y1 - b.y0 + R) / R;
    // (XCD column bands -- XCD k takes tile columns [k*n/8, (k+1)*n/8) of every strip, so that every
    // vertical re-read is a hit in ITS L2 -- cut the fabric reads to 1.02x compulsory and were slower:
    // 0.79-0.83 against 0.74-0.78 ms; removed)
    choose_block_shape(&nxw, &tpb, nxw >= 64 ? 4 : 0);   // wide rows: 4 waves per group measured best
    if (tpb > 8) tpb = 8;                               // launch bound of the kernel: 512 lanes
    const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
#define DLESM_XT(RR)                                                                                          \
    hipLaunchKernelGGL((jacobi5xt_tile<T, RR, DPP>), dim3(grid), dim3(64 * tpb), 0, s, in, out, ld, ny, b, cb, \
                       nxw, 0.25)
#ifdef DLESM_LAB
    switch (R) {
    case 2: if constexpr (T <= 4) DLESM_XT(2); break;
    case 4: if constexpr (T <= 4) DLESM_XT(4); break;
    case 6: if constexpr (T <= 4) DLESM_XT(6); break;
    case 12: if constexpr (T >= 4) DLESM_XT(12); break;
    case 16: if constexpr (T >= 4) DLESM_XT(16); break;    // (24 rows spill: 71-224 VGPRs at T = 6..8)
    default: DLESM_XT(8); break;
    }
#else
    DLESM_XT(best_rows[T]);
#endif
#undef DLESM_XT
}

// 1-based inclusive boxes: output box, last intermediate box, and per side (W,E,S,N) whether the
// earlier intermediate boxes grow by one cell per stage on that side
int launch_stencil5_multi(const double *in, double *out, int ld, int ny, int nsteps, int xstart, int xstop,
                          int ystart, int ystop, int exstart, int exstop, int eystart, int eystop, int gw,
                          int ge, int gs, int gn, hipStream_t s)
{
    // nsteps = 1 (internal callers only): the single step through this kernel's tile shape
    DLESM_REQUIRE(nsteps >= 1 && nsteps <= 8, "fused Jacobi steps: nsteps = %d (2..8 supported)", nsteps);
    DLESM_REQUIRE((gw | ge | gs | gn | 1) == 1, "fused Jacobi steps: grow flags must be 0 or 1");
    if (xstop < xstart || ystop < ystart) return DLESM_OK; // empty box: a zero-trip loop nest
    if (int rc = check_box("fused Jacobi steps", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && in != out, "fused Jacobi steps: null or aliased arrays");
    const bool empty_e = exstop < exstart || eystop < eystart;
    const int g = nsteps > 2 ? nsteps - 2 : 0;           // growth of the first intermediate box
    if (!empty_e)
        if (int rc = check_box("fused Jacobi steps (first intermediate box)", ld, ny, exstart - gw * g,
                               exstop + ge * g, eystart - gs * g, eystop + gn * g, 1))
            return rc;
    XtBoxes b{xstart - 1, xstop - 1, ystart - 1, ystop - 1, exstart - 1, exstop - 1, eystart - 1, eystop - 1,
              gw, ge, gs, gn};
    if (empty_e) { b.ex0 = b.ey0 = 1; b.ex1 = b.ey1 = 0; b.gw = b.ge = b.gs = b.gn = 0; }
    // 16-byte lanes need the last column any lane may read inside the last whole 2-column chunk
    // of a row, and 16-byte aligned bases
    const int last = std::max(b.x1, b.ex1 + b.ge * g) + 1;
    const bool vec2 = !(tuning("j5_variant", 0) & 4) && last <= 2 * (ld / 2) - 1 && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    if (!vec2 && nsteps != 2) {
        // Arrays that do not meet the 16-byte-lane conditions (odd leading dimension with the east ring column
        // in the last chunk, unaligned bases): no fused kernel -- the same result through nsteps single
        // sweeps and two stream-ordered scratch copies of the field, exactly the definition in the header:
        // t_0 = in; t_s = J(t_{s-1}) on the stage box E_s, t_{s-1} elsewhere; out = J(t_{nsteps-1}) on the box.
        const size_t bytes = (size_t)ld * ny * sizeof(double);
        double *ta = nullptr, *tb = nullptr;
        DLESM_HIP_TRY(hipMallocAsync((void **)&ta, bytes, s));
        if (hipMallocAsync((void **)&tb, bytes, s) != hipSuccess) {
            (void)hipFreeAsync(ta, s);
            return fail(DLESM_EHIP, "fused Jacobi steps (fallback): scratch allocation failed");
        }
        int rc = DLESM_OK;
        if (hipMemcpyAsync(ta, in, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = fail(DLESM_EHIP, "fallback copy failed");
        for (int st = 1; st <= nsteps - 1 && !rc; st++) {
            if (empty_e) break;                              // every stage box empty: t_s = in
            const int k = nsteps - 1 - st;                   // growth of stage box E_st
            if (hipMemcpyAsync(tb, ta, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = fail(DLESM_EHIP, "fallback copy failed"); break; }
            rc = launch_stencil5(ta, tb, ld, ny, exstart - gw * k, exstop + ge * k, eystart - gs * k, eystop + gn * k, s);
            std::swap(ta, tb);
        }
        if (!rc) rc = launch_stencil5(ta, out, ld, ny, xstart, xstop, ystart, ystop, s);
        (void)hipFreeAsync(ta, s);
        (void)hipFreeAsync(tb, s);
        return rc;
    }
    if (!vec2) {
        dim3 grid((unsigned)((b.x1 - b.x0 + 256) / 256), 1);
        for (int yb = b.y0; yb <= b.y1; yb += 65535) {   // grid.y is limited to 65535 rows per launch
            const int ye = std::min(b.y1, yb + 65534);
            grid.y = (unsigned)(ye - yb + 1);
            hipLaunchKernelGGL(jacobi5x2_direct, grid, dim3(256), 0, s, in, out, ld, b.x0, b.x1, yb, ye, b.ex0, b.ex1,
                               b.ey0, b.ey1);
        }
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    const int R = tuning("j5xt_rows", 0);
#ifdef DLESM_LAB
    const bool dpp = tuning("j5xt_dpp", 1);
    const bool march = tuning("j5xt_march", 0);
#define DLESM_T(TT)                                                                             \
    do {                                                                                        \
        if (dpp) {                                                                              \
            if (!(march && launch_xt_region<TT, true>(in, out, ld, ny, b, R, s)))               \
                launch_xt<TT, true>(in, out, ld, ny, b, R, s);                                  \
        } else {                                                                                \
            if (!(march && launch_xt_region<TT, false>(in, out, ld, ny, b, R, s)))              \
                launch_xt<TT, false>(in, out, ld, ny, b, R, s);                                 \
        }                                                                                       \
    } while (0)
#else
#define DLESM_T(TT) launch_xt<TT, true>(in, out, ld, ny, b, R, s)      // (DPP wave shifts, the tile form)
#endif
    switch (nsteps) {
#ifdef DLESM_LAB
    case 1: DLESM_T(1); break;                           // (j5_kernel = 3: the single step through this tile shape)
#endif
    case 2: DLESM_T(2); break;
    case 3: DLESM_T(3); break;
    case 4: DLESM_T(4); break;
    case 5: DLESM_T(5); break;
    case 6: DLESM_T(6); break;
    case 7: DLESM_T(7); break;
    default: DLESM_T(8); break;
    }
#undef DLESM_T
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
    return dlesm::launch_stencil5_multi(in, out, ld, ny, 2, xstart, xstop, ystart, ystop, exstart, exstop,
                                        eystart, eystop, 0, 0, 0, 0, (hipStream_t)stream);
}

extern "C" int dlesm_stencil5_multi_f64(const double *in, double *out, int ld, int ny, int nsteps, int xstart,
                                        int xstop, int ystart, int ystop, int exstart, int exstop,
                                        int eystart, int eystop, int grow_w, int grow_e, int grow_s,
                                        int grow_n, void *stream)
{
    dlesm::clear_error();
    if (nsteps < 2) return dlesm::fail(DLESM_EINVAL, "fused Jacobi steps: nsteps = %d (2..8 supported)", nsteps);
    if (int rc = dlesm::ensure_device()) return rc;
    return dlesm::launch_stencil5_multi(in, out, ld, ny, nsteps, xstart, xstop, ystart, ystop, exstart, exstop,
                                        eystart, eystop, grow_w, grow_e, grow_s, grow_n, (hipStream_t)stream);
}
