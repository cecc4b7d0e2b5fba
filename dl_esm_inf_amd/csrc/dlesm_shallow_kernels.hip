// The GOcean `shallow` kernel set as SEPARATE launch entries -- one per PSy loop nest, which is what
// an unmodified PSyclone-generated PSy layer has (one `do jj / do ji / call kern_code(ji, jj, ...)`
// nest per kernel; form: infrastructure_mod.f90:13-41, metadata argument_mod.f90:39-112,
// kernel_mod.f90:28-50):
//
//     compute_cu, compute_cv, compute_z, compute_h        (u, v, p          -> cu, cv, z, h)
//     compute_unew, compute_vnew, compute_pnew            (old level + cu, cv, z, h -> new level)
//     time_smooth                                         (Asselin filter of the old level)
//
// for both staggerings (index_offset GO_OFFSET_NE and GO_OFFSET_SW).  The formulas are those of
// DESIGN.md section 6 / 6.2 / 6.3 -- the same expression trees as the fused step (dlesm_shallow.hip)
// and the CPU checker, compiled with -ffp-contract=off: the seven launches produce, bit for bit,
// what the fused step produces.  The difference is traffic: 224 B/cell for the sequence (every
// intermediate goes through HBM) against 72 B/cell fused.
//
// Every kernel is one instance of the same wave-tile sweep the other kernels use: 64 lanes x 2
// doubles (16-byte lanes, 1 KiB of a row per wave) x R rows, tiles numbered row-major so that
// workgroups -- dispatched in index order -- sweep memory linearly.  West / east operands come from the
// neighbouring lane through a wave shift on the VALU; the one column a wave cannot get from its own lanes
// is fetched by lane 0 / lane 63 (one 8-byte load per row of the arrays that need it, an L1/L2 hit), so
// that all 64 lanes store and every wave tile covers whole 128-byte lines -- measured against tiles that
// give up a halo lane per side (62/63 output lanes, tile edges inside a line) this is 1-4 points faster per kernel
// (cu 69.0 -> 72.0 % of peak at 8192^2).
// South / north operands come from the extra row loaded below / above the tile.  A kernel is written ONCE, as an
// expression over an accessor `at<array, di, dj>()`: the tile sweep instantiates it on register rows
// (two columns at a time), the one-cell-per-thread form for odd leading dimensions on memory.
#include "dlesm_internal.h"

namespace dlesm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

// two adjacent columns of a row; the operators are the scalar operations applied to each column,
// in the order written (no contraction), so an expression over V2 is the scalar expression twice
struct V2 { double x, y; };
__device__ __forceinline__ V2 operator+(const V2 &a, const V2 &b) { return V2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ V2 operator-(const V2 &a, const V2 &b) { return V2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ V2 operator*(const V2 &a, const V2 &b) { return V2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ V2 operator/(const V2 &a, const V2 &b) { return V2{a.x / b.x, a.y / b.y}; }
__device__ __forceinline__ V2 operator*(double s, const V2 &b) { return V2{s * b.x, s * b.y}; }

struct KArgs {
    const double *in[4];
    double *out;
    double s0, s1;            // the kernel's real scalars
};

// ---- the kernels: out(i,j) = eval(at<array, di, dj>() ...) -----------------------------------------
// NIN input arrays; W/E/S/N: some operand lies one cell to that side (the box needs a ring there); EW[a]: array a is
// read one column west (bit 0) / east (bit 1) -- lanes 0 / 63 then fetch the column outside the wave; RS[a] / RN[a]:
// array a is read one row south / north.
#define AT(a, di, dj) t.template at<a, di, dj>()

struct CuNE {   // cu(i,j) = 0.5*(p(i+1,j)+p(i,j))*u(i,j)                     in: p, u
    static constexpr int NIN = 2;
    static constexpr bool W = false, E = true, S = false, N = false;
    static constexpr int EW[4] = {2, 0, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double, double) { return 0.5 * (AT(0, 1, 0) + AT(0, 0, 0)) * AT(1, 0, 0); }
};
struct CvNE {   // cv(i,j) = 0.5*(p(i,j+1)+p(i,j))*v(i,j)                     in: p, v
    static constexpr int NIN = 2;
    static constexpr bool W = false, E = false, S = false, N = true;
    static constexpr int EW[4] = {0, 0, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {1, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double, double) { return 0.5 * (AT(0, 0, 1) + AT(0, 0, 0)) * AT(1, 0, 0); }
};
struct ZNE {    // z(i,j) = (fsdx*(v(i+1,j)-v(i,j)) - fsdy*(u(i,j+1)-u(i,j))) / (p(i,j)+p(i+1,j)+p(i+1,j+1)+p(i,j+1))   in: p, u, v
    static constexpr int NIN = 3;
    static constexpr bool W = false, E = true, S = false, N = true;
    static constexpr int EW[4] = {2, 0, 2, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {1, 1, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double fsdx, double fsdy)
    {
        return (fsdx * (AT(2, 1, 0) - AT(2, 0, 0)) - fsdy * (AT(1, 0, 1) - AT(1, 0, 0))) /
               (AT(0, 0, 0) + AT(0, 1, 0) + AT(0, 1, 1) + AT(0, 0, 1));
    }
};
struct HNE {    // h(i,j) = p(i,j) + 0.25*(u(i,j)^2 + u(i-1,j)^2 + v(i,j)^2 + v(i,j-1)^2)            in: p, u, v
    static constexpr int NIN = 3;
    static constexpr bool W = true, E = false, S = true, N = false;
    static constexpr int EW[4] = {0, 1, 0, 0};
    static constexpr int RS[4] = {0, 0, 1, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double, double)
    {
        return AT(0, 0, 0) + 0.25 * (AT(1, 0, 0) * AT(1, 0, 0) + AT(1, -1, 0) * AT(1, -1, 0) + AT(2, 0, 0) * AT(2, 0, 0) +
                                     AT(2, 0, -1) * AT(2, 0, -1));
    }
};
struct UnewNE { // unew = uold + tdts8*(z(i,j)+z(i,j-1))*(cv(i+1,j)+cv(i,j)+cv(i,j-1)+cv(i+1,j-1)) - tdtsdx*(h(i+1,j)-h(i,j))   in: uold, z, cv, h
    static constexpr int NIN = 4;
    static constexpr bool W = false, E = true, S = true, N = false;
    static constexpr int EW[4] = {0, 0, 2, 2};
    static constexpr int RS[4] = {0, 1, 1, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double tdts8, double tdtsdx)
    {
        return AT(0, 0, 0) + tdts8 * (AT(1, 0, 0) + AT(1, 0, -1)) * (AT(2, 1, 0) + AT(2, 0, 0) + AT(2, 0, -1) + AT(2, 1, -1)) -
               tdtsdx * (AT(3, 1, 0) - AT(3, 0, 0));
    }
};
struct VnewNE { // vnew = vold - tdts8*(z(i,j)+z(i-1,j))*(cu(i,j+1)+cu(i-1,j+1)+cu(i-1,j)+cu(i,j)) - tdtsdy*(h(i,j+1)-h(i,j))   in: vold, z, cu, h
    static constexpr int NIN = 4;
    static constexpr bool W = true, E = false, S = false, N = true;
    static constexpr int EW[4] = {0, 1, 1, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 1, 1};
    template <class A> __device__ static auto eval(const A &t, double tdts8, double tdtsdy)
    {
        return AT(0, 0, 0) - tdts8 * (AT(1, 0, 0) + AT(1, -1, 0)) * (AT(2, 0, 1) + AT(2, -1, 1) + AT(2, -1, 0) + AT(2, 0, 0)) -
               tdtsdy * (AT(3, 0, 1) - AT(3, 0, 0));
    }
};
struct PnewNE { // pnew = pold - tdtsdx*(cu(i,j)-cu(i-1,j)) - tdtsdy*(cv(i,j)-cv(i,j-1))             in: pold, cu, cv
    static constexpr int NIN = 3;
    static constexpr bool W = true, E = false, S = true, N = false;
    static constexpr int EW[4] = {0, 1, 0, 0};
    static constexpr int RS[4] = {0, 0, 1, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double tdtsdx, double tdtsdy)
    {
        return AT(0, 0, 0) - tdtsdx * (AT(1, 0, 0) - AT(1, -1, 0)) - tdtsdy * (AT(2, 0, 0) - AT(2, 0, -1));
    }
};

// SW offset: the mirror image, with its own association order
struct CuSW {   // cu(i,j) = 0.5*(p(i,j)+p(i-1,j))*u(i,j)
    static constexpr int NIN = 2;
    static constexpr bool W = true, E = false, S = false, N = false;
    static constexpr int EW[4] = {1, 0, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double, double) { return 0.5 * (AT(0, 0, 0) + AT(0, -1, 0)) * AT(1, 0, 0); }
};
struct CvSW {   // cv(i,j) = 0.5*(p(i,j)+p(i,j-1))*v(i,j)
    static constexpr int NIN = 2;
    static constexpr bool W = false, E = false, S = true, N = false;
    static constexpr int EW[4] = {0, 0, 0, 0};
    static constexpr int RS[4] = {1, 0, 0, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double, double) { return 0.5 * (AT(0, 0, 0) + AT(0, 0, -1)) * AT(1, 0, 0); }
};
struct ZSW {    // z(i,j) = (fsdx*(v(i,j)-v(i-1,j)) - fsdy*(u(i,j)-u(i,j-1))) / (p(i-1,j-1)+p(i,j-1)+p(i,j)+p(i-1,j))
    static constexpr int NIN = 3;
    static constexpr bool W = true, E = false, S = true, N = false;
    static constexpr int EW[4] = {1, 0, 1, 0};
    static constexpr int RS[4] = {1, 1, 0, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double fsdx, double fsdy)
    {
        return (fsdx * (AT(2, 0, 0) - AT(2, -1, 0)) - fsdy * (AT(1, 0, 0) - AT(1, 0, -1))) /
               (AT(0, -1, -1) + AT(0, 0, -1) + AT(0, 0, 0) + AT(0, -1, 0));
    }
};
struct HSW {    // h(i,j) = p(i,j) + 0.25*(u(i+1,j)^2 + u(i,j)^2 + v(i,j+1)^2 + v(i,j)^2)
    static constexpr int NIN = 3;
    static constexpr bool W = false, E = true, S = false, N = true;
    static constexpr int EW[4] = {0, 2, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 1, 0};
    template <class A> __device__ static auto eval(const A &t, double, double)
    {
        return AT(0, 0, 0) + 0.25 * (AT(1, 1, 0) * AT(1, 1, 0) + AT(1, 0, 0) * AT(1, 0, 0) + AT(2, 0, 1) * AT(2, 0, 1) +
                                     AT(2, 0, 0) * AT(2, 0, 0));
    }
};
struct UnewSW { // unew = uold + tdts8*(z(i,j+1)+z(i,j))*(cv(i,j+1)+cv(i-1,j+1)+cv(i-1,j)+cv(i,j)) - tdtsdx*(h(i,j)-h(i-1,j))
    static constexpr int NIN = 4;
    static constexpr bool W = true, E = false, S = false, N = true;
    static constexpr int EW[4] = {0, 0, 1, 1};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 1, 1, 0};
    template <class A> __device__ static auto eval(const A &t, double tdts8, double tdtsdx)
    {
        return AT(0, 0, 0) + tdts8 * (AT(1, 0, 1) + AT(1, 0, 0)) * (AT(2, 0, 1) + AT(2, -1, 1) + AT(2, -1, 0) + AT(2, 0, 0)) -
               tdtsdx * (AT(3, 0, 0) - AT(3, -1, 0));
    }
};
struct VnewSW { // vnew = vold - tdts8*(z(i+1,j)+z(i,j))*(cu(i+1,j)+cu(i,j)+cu(i,j-1)+cu(i+1,j-1)) - tdtsdy*(h(i,j)-h(i,j-1))
    static constexpr int NIN = 4;
    static constexpr bool W = false, E = true, S = true, N = false;
    static constexpr int EW[4] = {0, 2, 2, 0};
    static constexpr int RS[4] = {0, 0, 1, 1}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double tdts8, double tdtsdy)
    {
        return AT(0, 0, 0) - tdts8 * (AT(1, 1, 0) + AT(1, 0, 0)) * (AT(2, 1, 0) + AT(2, 0, 0) + AT(2, 0, -1) + AT(2, 1, -1)) -
               tdtsdy * (AT(3, 0, 0) - AT(3, 0, -1));
    }
};
struct PnewSW { // pnew = pold - tdtsdx*(cu(i+1,j)-cu(i,j)) - tdtsdy*(cv(i,j+1)-cv(i,j))
    static constexpr int NIN = 3;
    static constexpr bool W = false, E = true, S = false, N = true;
    static constexpr int EW[4] = {0, 2, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 1, 0};
    template <class A> __device__ static auto eval(const A &t, double tdtsdx, double tdtsdy)
    {
        return AT(0, 0, 0) - tdtsdx * (AT(1, 1, 0) - AT(1, 0, 0)) - tdtsdy * (AT(2, 0, 1) - AT(2, 0, 0));
    }
};
// time_smooth (any offset):
//   field_old(i,j) = field(i,j) + alpha*(field_new(i,j) - 2.0*field(i,j) + field_old(i,j))      in: field, field_new, field_old
struct TimeSmooth {
    static constexpr int NIN = 3;
    static constexpr bool W = false, E = false, S = false, N = false;
    static constexpr int EW[4] = {0, 0, 0, 0};
    static constexpr int RS[4] = {0, 0, 0, 0}, RN[4] = {0, 0, 0, 0};
    template <class A> __device__ static auto eval(const A &t, double alpha, double)
    {
        return AT(0, 0, 0) + alpha * (AT(1, 0, 0) - 2.0 * AT(0, 0, 0) + AT(2, 0, 0));
    }
};
#undef AT

// ---- accessor over the register rows of a wave tile: row index k = row jb-1+k -----------------------
// edge[a][k]: the value of array a in row k one column WEST of lane 0's chunk (held by lane 0) or one column EAST
// of lane 63's chunk (held by lane 63) -- what the wave shift cannot deliver to those two lanes
template <int NIN, int R>
struct TileAcc {
    const V2 (&rows)[NIN][R + 2];
    const double (&edge)[NIN][R + 2];
    int k, lane;
    template <int a, int di, int dj> __device__ __forceinline__ V2 at() const
    {
        const V2 &r = rows[a][k + dj];
        if constexpr (di == 0) return r;
        else if constexpr (di > 0) {       // (i+1): own east column, the next lane's west one
            const double up = from_upper<true>(r.x);
            return V2{r.y, lane == 63 ? edge[a][k + dj] : up};
        } else {                           // (i-1)
            const double lo = from_lower<true>(r.y);
            return V2{lane == 0 ? edge[a][k + dj] : lo, r.x};
        }
    }
};
// ---- accessor over memory, one cell ----------------------------------------------------------------
struct CellAcc {
    const KArgs &a;
    size_t o;
    int ld;
    template <int k, int di, int dj> __device__ __forceinline__ double at() const { return a.in[k][o + di + (long)dj * ld]; }
};

// NT: non-temporal stores of the output; NTL: non-temporal loads of the arrays this kernel reads exactly once per cell
// (no south / north / west / east operand: uold in compute_unew, all three arrays of time_smooth ...)
template <class K, int R, bool NT, bool NTL = false>
__global__ __launch_bounds__(512) void swk_tile(KArgs a, int ld, int x0, int x1, int y0, int y1, int cb, int nxw)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int xw = w % nxw, strip = w / nxw;
    const int jb = y0 + strip * R;
    if (jb > y1) return;
    int je = jb + R - 1;
    if (je > y1) je = y1;
    const int c = cb + xw * 64 + lane;                 // this lane's chunk (2 columns)
    if (c - lane > x1 / 2) return;                     // idle padding tile
    const int c_ld = ld / 2 - 1;
    const int cl = c > c_ld ? c_ld : c;                // trailing lanes: any valid chunk
    const bool m0 = c <= c_ld && 2 * c >= x0 && 2 * c <= x1;
    const bool m1 = c <= c_ld && 2 * c + 1 >= x0 && 2 * c + 1 <= x1;
    const size_t col = (size_t)cl * 2;
    // the column outside the wave: west of lane 0 / east of lane 63 (clamped into the row; a clamped value is only
    // ever combined into cells outside the box)
    const int ecol = lane == 0 ? (2 * c - 1 < 0 ? 0 : 2 * c - 1) : (2 * c + 2 > ld - 1 ? ld - 1 : 2 * c + 2);

    V2 rows[K::NIN][R + 2];
    double edge[K::NIN][R + 2];
#pragma unroll
    for (int n = 0; n < K::NIN; n++) {
#pragma unroll
        for (int k = 1 - K::RS[n]; k <= R + K::RN[n]; k++) {
            int jj = jb - 1 + k;
            if (jj > je + K::RN[n]) jj = je + K::RN[n];
            const bool once = NTL && K::RS[n] == 0 && K::RN[n] == 0 && K::EW[n] == 0;
            const d2 v = once ? __builtin_nontemporal_load((const d2 *)(a.in[n] + (size_t)jj * ld + col))
                              : *(const d2 *)(a.in[n] + (size_t)jj * ld + col);
            rows[n][k] = V2{v.x, v.y};
            edge[n][k] = 0.0;
            if ((K::EW[n] & 1) && lane == 0) edge[n][k] = a.in[n][(size_t)jj * ld + ecol];
            if ((K::EW[n] & 2) && lane == 63) edge[n][k] = a.in[n][(size_t)jj * ld + ecol];
        }
    }
#pragma unroll
    for (int k = 1; k <= R; k++) {
        const int jj = jb - 1 + k;
        if (jj > je) break;
        const V2 r = K::eval(TileAcc<K::NIN, R>{rows, edge, k, lane}, a.s0, a.s1);
        double *po = a.out + (size_t)jj * ld + (size_t)c * 2;
        if (m0 && m1) {
            if constexpr (NT) __builtin_nontemporal_store(d2{r.x, r.y}, (d2 *)po);
            else *(d2 *)po = d2{r.x, r.y};
        } else {
            if (m0) po[0] = r.x;
            if (m1) po[1] = r.y;
        }
    }
}

// one cell per thread, operands from memory: odd leading dimensions, unaligned bases, thin boxes
template <class K>
__global__ __launch_bounds__(256) void swk_direct(KArgs a, int ld, int x0, int x1, int y0, int y1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > x1) return;
    for (int j = y0 + blockIdx.y; j <= y1; j += gridDim.y) {
        const size_t o = (size_t)j * ld + i;
        a.out[o] = K::eval(CellAcc{a, o, ld}, a.s0, a.s1);
    }
}

template <class K>
int launch_kernel(const char *who, const KArgs &a, int ld, int ny, int xstart, int xstop, int ystart, int ystop, void *stream)
{
    if (int rc = ensure_device()) return rc;
    if (xstop < xstart || ystop < ystart) return DLESM_OK;     // empty box: a zero-trip loop nest
    DLESM_REQUIRE(a.out != nullptr, "%s: null pointer", who);
    for (int n = 0; n < K::NIN; n++) DLESM_REQUIRE(a.in[n] != nullptr, "%s: null pointer", who);
    if (ld < 1 || ny < 1) return fail(DLESM_EINVAL, "%s: array extents %dx%d", who, ld, ny);
    if (xstart - (K::W ? 1 : 0) < 1 || xstop + (K::E ? 1 : 0) > ld || ystart - (K::S ? 1 : 0) < 1 || ystop + (K::N ? 1 : 0) > ny)
        return fail(DLESM_EINVAL, "%s: box (%d:%d,%d:%d) plus the cells its stencil reads (W%d E%d S%d N%d) does not fit "
                                  "in an array of %dx%d", who, xstart, xstop, ystart, ystop, (int)K::W, (int)K::E, (int)K::S,
                    (int)K::N, ld, ny);
    // the output may alias an input only where that input is read at (i,j) alone
    for (int n = 0; n < K::NIN; n++)
        if ((const double *)a.out == a.in[n])
            DLESM_REQUIRE(!(K::W || K::E || K::S || K::N), "%s: the output aliases an input read at neighbouring points", who);
    hipStream_t s = (hipStream_t)stream;
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    // 16-byte lanes; on an odd leading dimension (rows alternately 8-byte aligned) only while every column read
    // stays inside the last whole 2-column chunk of a row
    bool aligned = ld % 2 == 0 || x1 + (K::E ? 1 : 0) <= 2 * (ld / 2) - 1;
    aligned = aligned && (uintptr_t)a.out % 16 == 0;
    for (int n = 0; n < K::NIN; n++) aligned = aligned && (uintptr_t)a.in[n] % 16 == 0;
    const int nx = x1 - x0 + 1, h = y1 - y0 + 1;
    const bool thin = nx <= SW_THIN_BOX && h > 8;
    if (aligned && !thin && tuning("swk_kernel", 0) == 0) {
        constexpr int R = 2;
        const int cb = (x0 / 2) & ~7, c_last = x1 / 2;        // tiles anchored on 128-byte lines of the row
        int nxw = (c_last - cb + 64) / 64, tpb = 4;
        // 4 waves per group, a quarter of a group past / short of a multiple of 8 groups per row: measured at 8192^2
        // (scripts/shallow_r3_probe.py --what kernels,shapes: 65 tiles per row) every kernel of the set is fastest at
        // 4 waves x 65 tiles (16.25 groups per row), 1-2 points above 8 waves x 66 tiles, the Jacobi rule's choice
        choose_block_shape(&nxw, &tpb, 4);
        if (tpb > 8) tpb = 8;                                  // the kernel is bounded to 512 threads
        const int strips = (h + R - 1) / R;
        const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
        // Store policy: as the other sweeps, non-temporal once the arrays outgrow the Infinity Cache.  A kernel that updates
        // an array IN PLACE (time_smooth) writes the line it has just read: with ordinary loads that line sits in L2 when the
        // store arrives, and the kernel ran at 66-68 % of peak whatever the store policy; with non-temporal loads AND stores
        // 75.4 % (8192^2, one process, scripts/shallow_r3_probe.py) -- a linear in-place sweep of the same arrays reaches
        // 71.3-71.9 %, scripts/inplace_probe.py.  For the other kernels non-temporal loads of the once-read arrays change
        // nothing (74.3 against 74.3 %), so only in-place kernels take them.
        bool in_place = false;
        for (int n = 0; n < K::NIN; n++) in_place = in_place || (const double *)a.out == a.in[n];
        const int key = tuning("swk_nt", -1), lkey = tuning("swk_ntl", -1);
        const bool nts = key >= 0 ? key != 0 : nt_stores_for(ld, y0, y1) != 0;
        const bool ntl = lkey >= 0 ? lkey != 0 : (in_place && nts);
        if (nts && ntl) hipLaunchKernelGGL((swk_tile<K, R, true, true>), dim3(grid), dim3(64 * tpb), 0, s, a, ld, x0, x1, y0, y1, cb, nxw);
        else if (nts) hipLaunchKernelGGL((swk_tile<K, R, true>), dim3(grid), dim3(64 * tpb), 0, s, a, ld, x0, x1, y0, y1, cb, nxw);
        else if (ntl) hipLaunchKernelGGL((swk_tile<K, R, false, true>), dim3(grid), dim3(64 * tpb), 0, s, a, ld, x0, x1, y0, y1, cb, nxw);
        else hipLaunchKernelGGL((swk_tile<K, R, false>), dim3(grid), dim3(64 * tpb), 0, s, a, ld, x0, x1, y0, y1, cb, nxw);
    } else {
        hipLaunchKernelGGL((swk_direct<K>), dim3((nx + 255) / 256, h > 4096 ? 4096 : h), dim3(256), 0, s, a, ld, x0, x1, y0, y1);
    }
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

inline int bad_offset(const char *who, int offset)
{
    return fail(DLESM_EINVAL, "%s: index offset %d is neither DLESM_OFFSET_NE nor DLESM_OFFSET_SW", who, offset);
}

} // namespace

} // namespace dlesm

using namespace dlesm;

#define DLESM_BY_OFFSET(who, KNE, KSW)                                                                           \
    do {                                                                                                         \
        if (offset == DLESM_OFFSET_NE) return launch_kernel<KNE>(who, a, ld, ny, xstart, xstop, ystart, ystop, stream); \
        if (offset == DLESM_OFFSET_SW) return launch_kernel<KSW>(who, a, ld, ny, xstart, xstop, ystart, ystop, stream); \
        return bad_offset(who, offset);                                                                          \
    } while (0)

extern "C" int dlesm_compute_cu_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double *cu,
                                    const double *p, const double *u, void *stream)
{
    const KArgs a{{p, u, nullptr, nullptr}, cu, 0.0, 0.0};
    DLESM_BY_OFFSET("dlesm_compute_cu_f64", CuNE, CuSW);
}

extern "C" int dlesm_compute_cv_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double *cv,
                                    const double *p, const double *v, void *stream)
{
    const KArgs a{{p, v, nullptr, nullptr}, cv, 0.0, 0.0};
    DLESM_BY_OFFSET("dlesm_compute_cv_f64", CvNE, CvSW);
}

extern "C" int dlesm_compute_z_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double fsdx,
                                   double fsdy, double *z, const double *p, const double *u, const double *v, void *stream)
{
    const KArgs a{{p, u, v, nullptr}, z, fsdx, fsdy};
    DLESM_BY_OFFSET("dlesm_compute_z_f64", ZNE, ZSW);
}

extern "C" int dlesm_compute_h_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double *h,
                                   const double *p, const double *u, const double *v, void *stream)
{
    const KArgs a{{p, u, v, nullptr}, h, 0.0, 0.0};
    DLESM_BY_OFFSET("dlesm_compute_h_f64", HNE, HSW);
}

extern "C" int dlesm_compute_unew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double tdts8,
                                      double tdtsdx, double *unew, const double *uold, const double *z, const double *cv,
                                      const double *h, void *stream)
{
    const KArgs a{{uold, z, cv, h}, unew, tdts8, tdtsdx};
    DLESM_BY_OFFSET("dlesm_compute_unew_f64", UnewNE, UnewSW);
}

extern "C" int dlesm_compute_vnew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double tdts8,
                                      double tdtsdy, double *vnew, const double *vold, const double *z, const double *cu,
                                      const double *h, void *stream)
{
    const KArgs a{{vold, z, cu, h}, vnew, tdts8, tdtsdy};
    DLESM_BY_OFFSET("dlesm_compute_vnew_f64", VnewNE, VnewSW);
}

extern "C" int dlesm_compute_pnew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop, double tdtsdx,
                                      double tdtsdy, double *pnew, const double *pold, const double *cu, const double *cv,
                                      void *stream)
{
    const KArgs a{{pold, cu, cv, nullptr}, pnew, tdtsdx, tdtsdy};
    DLESM_BY_OFFSET("dlesm_compute_pnew_f64", PnewNE, PnewSW);
}

extern "C" int dlesm_time_smooth_f64(int ld, int ny, int xstart, int xstop, int ystart, int ystop, double alpha,
                                     const double *field, const double *field_new, double *field_old, void *stream)
{
    const KArgs a{{field, field_new, field_old, nullptr}, field_old, alpha, 0.0};
    return launch_kernel<TimeSmooth>("dlesm_time_smooth_f64", a, ld, ny, xstart, xstop, ystart, ystop, stream);
}
