// Device-side helpers shared by the kernels of several translation units (frame numbering, the
// one-cell shallow-water update).  Included after dlesm_internal.h.
#ifndef DLESM_DEVICE_H
#define DLESM_DEVICE_H

#include "dlesm_internal.h"

namespace dlesm {

// One cell of the one-cell-wide frame of the box (x0:x1, y0:y1), numbered t = 0 .. frame_cells-1:
// south row, north row, then the west and east columns between them.  Cells of a west/east column
// that a neighbour will receive also go into their send-buffer slot, in the j order of the pack
// loop (parallel_comms_mod.f90:1678-1683).
__device__ __forceinline__ long frame_cells(int w, int h)
{
    const int ncol = h > 2 ? h - 2 : 0, nrows = h > 1 ? 2 : 1;
    return (long)nrows * w + 2L * ncol * (w > 1 ? 1 : 0) + (w == 1 ? ncol : 0);
}

// frame cell number t -> (i, j): south row, north row, then the west and east columns between them
__device__ __forceinline__ void frame_index(long t, int x0, int x1, int y0, int y1, int &i, int &j)
{
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    const int ncol = h > 2 ? h - 2 : 0, nrows = h > 1 ? 2 : 1;
    if (t < (long)nrows * w) {
        j = t < w ? y0 : y1;
        i = x0 + (int)(t % w);
    } else {
        long k = t - (long)nrows * w;
        if (w == 1) { i = x0; j = y0 + 1 + (int)k; }
        else { i = k < ncol ? x0 : x1; j = y0 + 1 + (int)(k % ncol); }
    }
}

// ---- hand-overs between workgroups that are running at the same time ------------------------------------------------
// The two halves of the form MI355X_MICROARCH.md lists under "Valid forms" (per-XCD L2s are not coherent with each
// other, a CU's L1 is never refreshed by another CU's stores).  SYSTEM: the bytes were written by, or are meant for,
// another GPU (mailboxes); otherwise agent scope.
//   consumer: ONE lane polls with relaxed loads; when the poll has matched it calls handover_acquire() -- the acquire
//             fence (buffer_inv) and the s_waitcnt that waits for the invalidate to have completed -- and the workgroup
//             then passes __syncthreads() before any lane loads the handed-over bytes;
//   producer: every storing wave `s_waitcnt vmcnt(0)`, __syncthreads(), then ONE lane calls handover_release() -- the
//             release fence (buffer_wbl2) and an s_waitcnt written as inline asm, which the compiler cannot drop as it
//             does the fence's own wait when it believes the lane's vmcnt scoreboard empty (the guide's "Compiler
//             hazard": our publishing lane has just used a returned atomic) -- and only then stores the flag, relaxed.
template <bool SYSTEM>
__device__ __forceinline__ void handover_acquire()
{
    if constexpr (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <bool SYSTEM>
__device__ __forceinline__ void handover_release()
{
    if constexpr (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- a box swept as ROW SEGMENTS (the utility kernels: fill, initial condition, checksum, gather copies) ----------
// One workgroup (256 threads) = one segment of one row of the box: up to SEG_PAIRS 16-byte pairs, four per thread.  The
// pairs of a row are anchored on the 128-BYTE LINE that holds the row's first element (elements of the line that lie
// before the box are masked), so that every wave's 1 KiB access covers whole lines -- a sweep whose chunks start 16
// bytes into a line measured 51 % of the HBM peak as a pure store stream against 85 % aligned.  Work items are numbered
// row-major and are short-lived: workgroups, dispatched in index order, sweep memory front to back like the linear copy
// that sets the measured ceiling.  Needs a 16-byte aligned base and rows of at least ROWSEG_MIN_NX elements.
constexpr int SEG_PAIRS = 1024;
constexpr int ROWSEG_MIN_NX = 4;
typedef double rs_d2 __attribute__((ext_vector_type(2)));
typedef double rs_d2a8 __attribute__((ext_vector_type(2), aligned(8)));

// a row of nx doubles as `segs` equal segments of `segp` pairs (a multiple of 64, at most `cap` <= SEG_PAIRS): no nearly
// empty last segment.  Worst case pairs per row: the 15 masked elements of the first line + nx, rounded up.
inline void rowseg_split(int nx, int cap, int *segs, int *segp)
{
    if (cap < 64 || cap > SEG_PAIRS) cap = SEG_PAIRS;
    const int npairs = (nx + 15) / 2 + 1;
    *segs = (npairs + cap - 1) / cap;
    *segp = (((npairs + *segs - 1) / *segs) + 63) & ~63;
}

// What thread `tid` of segment `sg` does with its k-th pair: el = element index (from the array base) of the pair's
// first element, m0 / m1 = which of the two elements lie in [e0, e1] (the row's part of the box)
struct RowPair {
    long el;
    bool m0, m1;
    __device__ __forceinline__ bool any() const { return m0 || m1; }
    __device__ __forceinline__ bool full() const { return m0 && m1; }
};
__device__ __forceinline__ RowPair rowseg_pair(long e0, long e1, int sg, int segp, int tid, int k)
{
    const int q = tid + 256 * k;
    const long el = (e0 & ~15L) + 2L * ((long)sg * segp + q);
    const bool in = q < segp;
    return RowPair{el, in && el >= e0 && el <= e1, in && el + 1 >= e0 && el + 1 <= e1};
}

struct SwPoint { double un, vn, pn; };

// the NE-offset update of ONE cell, operands straight from memory: the
// expression trees of the oracle's compute_*_code.  DEV: the current fields are read at device
// scope (past this XCD's L2) -- halo cells an exchange may have written after this kernel started.
template <bool DEV = false>
__device__ __forceinline__ SwPoint shallow_values_ne(
    const dlesm_sw_params &q, int ld, size_t o, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ p, const double *__restrict__ uold, const double *__restrict__ vold,
    const double *__restrict__ pold)
{
    auto get = [](const double *ptr) {
        if constexpr (DEV) return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *ptr;
    };
#define U_(di, dj) get(u + (o + (di) + (long)(dj) * ld))
#define V_(di, dj) get(v + (o + (di) + (long)(dj) * ld))
#define P_(di, dj) get(p + (o + (di) + (long)(dj) * ld))
    // cu(a,b) = 0.5*(p(a+1,b)+p(a,b))*u(a,b) ; cv(a,b) = 0.5*(p(a,b+1)+p(a,b))*v(a,b)
#define CU(di, dj) (0.5 * (P_((di) + 1, dj) + P_(di, dj)) * U_(di, dj))
#define CV(di, dj) (0.5 * (P_(di, (dj) + 1) + P_(di, dj)) * V_(di, dj))
    // z(a,b) at the NE corner of T(a,b)
#define Z(di, dj)                                                                                   \
    ((q.fsdx * (V_((di) + 1, dj) - V_(di, dj)) - q.fsdy * (U_(di, (dj) + 1) - U_(di, dj))) /         \
     (P_(di, dj) + P_((di) + 1, dj) + P_((di) + 1, (dj) + 1) + P_(di, (dj) + 1)))
#define H(di, dj)                                                                                   \
    (P_(di, dj) + 0.25 * (U_(di, dj) * U_(di, dj) + U_((di)-1, dj) * U_((di)-1, dj) +                \
                          V_(di, dj) * V_(di, dj) + V_(di, (dj)-1) * V_(di, (dj)-1)))
    const double z00 = Z(0, 0), z0m = Z(0, -1), zm0 = Z(-1, 0);
    const double h00 = H(0, 0), hp0 = H(1, 0), h0p = H(0, 1);
    const double cu00 = CU(0, 0), cum0 = CU(-1, 0), cu0p = CU(0, 1), cump = CU(-1, 1);
    const double cv00 = CV(0, 0), cv0m = CV(0, -1), cvp0 = CV(1, 0), cvpm = CV(1, -1);
    SwPoint r;
    r.un = uold[o] + q.tdts8 * (z00 + z0m) * (cvp0 + cv00 + cv0m + cvpm) - q.tdtsdx * (hp0 - h00);
    r.vn = vold[o] - q.tdts8 * (z00 + zm0) * (cu0p + cump + cum0 + cu00) - q.tdtsdy * (h0p - h00);
    r.pn = pold[o] - q.tdtsdx * (cu00 - cum0) - q.tdtsdy * (cv00 - cv0m);
#undef U_
#undef V_
#undef P_
#undef CU
#undef CV
#undef Z
#undef H
    return r;
}

// time_smooth of ONE cell of the old level, in place: field_old = field + alpha*(field_new -
// 2*field + field_old) for u, v and p, r holding the new level of the cell
__device__ __forceinline__ void smooth_old_level(double alpha, size_t o, const double *u, const double *v, const double *p,
                                                 const SwPoint &r, double *uold, double *vold, double *pold)
{
    uold[o] = u[o] + alpha * (r.un - 2.0 * u[o] + uold[o]);
    vold[o] = v[o] + alpha * (r.vn - 2.0 * v[o] + vold[o]);
    pold[o] = p[o] + alpha * (r.pn - 2.0 * p[o] + pold[o]);
}

__device__ __forceinline__ void shallow_point_ne(
    const dlesm_sw_params &q, int ld, size_t o, const double *__restrict__ u, const double *__restrict__ v,
    const double *__restrict__ p, const double *__restrict__ uold, const double *__restrict__ vold,
    const double *__restrict__ pold, double *__restrict__ unew, double *__restrict__ vnew, double *__restrict__ pnew)
{
    const SwPoint r = shallow_values_ne(q, ld, o, u, v, p, uold, vold, pold);
    unew[o] = r.un;
    vnew[o] = r.vn;
    pnew[o] = r.pn;
}

} // namespace dlesm

#endif
