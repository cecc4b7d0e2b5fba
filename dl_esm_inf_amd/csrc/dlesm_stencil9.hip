// A general 9-point (3 x 3) weighted stencil over an r2d_field: the PSy-layer loop nest of any
// GOcean kernel of the form
//     out(ji,jj) = SUM_{dj=-1..1} SUM_{di=-1..1} c(di,dj) * in(ji+di, jj+dj)
// (metadata: go_arg(GO_WRITE, GO_CT, GO_POINTWISE), go_arg(GO_READ, GO_CT, GO_STENCIL(111,111,111)),
// nine GO_R_SCALAR coefficients; argument_mod.f90:39-112).  Five-point kernels are the same entry
// with zero corner weights.  The reference has no stencil loop: the evaluation
// order is frozen in DESIGN.md section 5.9 --
//     S = (c_sw*in(i-1,j-1) + c_s*in(i,j-1)) + c_se*in(i+1,j-1)
//     M = (c_w *in(i-1,j  ) + c_c*in(i,j  )) + c_e *in(i+1,j  )
//     N = (c_nw*in(i-1,j+1) + c_n*in(i,j+1)) + c_ne*in(i+1,j+1)
//     out = (S + M) + N
// every product and sum rounded (FMA contraction off), so that GPU and CPU agree bit for bit.
//
// 16 B/cell of algorithmic traffic, exactly as the Jacobi sweep, and the same kernel shape: wave
// tiles of 64 lanes x 2 columns x 2 rows swept linearly, the four rows a tile needs loaded up front,
// west/east values of all three stencil rows from the neighbouring lane, lanes 0 and 63 fetching the
// one column outside the wave.
#include "dlesm_internal.h"
#include "dlesm_device.h"

namespace dlesm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

struct Coef9 { double sw, s, se, w, c, e, nw, n, ne; };

__device__ __forceinline__ double point9(const Coef9 &k, double sw, double s, double se, double w, double c, double e,
                                         double nw, double n, double ne)
{
    const double S = (k.sw * sw + k.s * s) + k.se * se;
    const double M = (k.w * w + k.c * c) + k.e * e;
    const double N = (k.nw * nw + k.n * n) + k.ne * ne;
    return (S + M) + N;
}

constexpr int R = 2;

template <bool NTS>
__device__ __forceinline__ void stencil9_tile_body(const double *__restrict__ in, double *__restrict__ out, const Coef9 &k,
                                                   int ld, int x0, int x1, int y0, int y1, int c_first, int nxw,
                                                   unsigned block)
{
    const int lane = threadIdx.x & 63;
    const int w = block * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
    double ev[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
        int jj = jb - 1 + r;
        if (jj > je + 1) jj = je + 1;
        const size_t o = (size_t)jj * ld;
        v[r] = *(const d2 *)(in + o + (size_t)cl * 2);
        ev[r] = ecol >= 0 ? in[o + ecol] : 0.0;
    }
    // west neighbour of column .x and east neighbour of column .y, for every loaded row
    double vw[R + 2], ve[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
        vw[r] = from_lower<true>(v[r].y);        // whole-wave shifts on the VALU (DPP), not through the LDS pipe
        ve[r] = from_upper<true>(v[r].x);
        if (lane == 0) vw[r] = ev[r];
        if (lane == 63) ve[r] = ev[r];
    }
#pragma unroll
    for (int r = 1; r <= R; r++) {
        if (jb + r - 1 > je) break;
        const double o0 = point9(k, vw[r - 1], v[r - 1].x, v[r - 1].y, vw[r], v[r].x, v[r].y, vw[r + 1], v[r + 1].x, v[r + 1].y);
        const double o1 = point9(k, v[r - 1].x, v[r - 1].y, ve[r - 1], v[r].x, v[r].y, ve[r], v[r + 1].x, v[r + 1].y, ve[r + 1]);
        double *po = out + (size_t)(jb + r - 1) * ld + (size_t)c * 2;
        if (m0 && m1) {
            if (NTS) __builtin_nontemporal_store(d2{o0, o1}, (d2 *)po);
            else *(d2 *)po = d2{o0, o1};
        } else {
            if (m0) po[0] = o0;
            if (m1) po[1] = o1;
        }
    }
}

template <bool NTS>
__global__ __launch_bounds__(1024) void stencil9_tile(const double *__restrict__ in, double *__restrict__ out,
                                                     Coef9 k, int ld, int x0, int x1, int y0, int y1, int c_first,
                                                     int nxw)
{
    stencil9_tile_body<NTS>(in, out, k, ld, x0, x1, y0, y1, c_first, nxw, blockIdx.x);
}

// The distributed step in ONE launch (the construction of jacobi5_tile_framed): the first fj.nblocks
// workgroups compute the one-cell frame of the box (fj.fx0:fx1, fj.fy0:fy1) from memory, store it write-through
// at device scope -- into `out` and, for the west/east columns, into the send buffer -- and the last of them
// publishes fj.seq in the flag the side stream's waiter sleeps on; all other workgroups are the ordinary tile
// sweep over the interior (x0:x1, y0:y1).  Joined form only: `in` holds valid halos when the launch starts.
template <bool NTS>
__global__ __launch_bounds__(1024) void stencil9_tile_framed(const double *__restrict__ in, double *__restrict__ out,
                                                            Coef9 k, int ld, int x0, int x1, int y0, int y1, int c_first,
                                                            int nxw, FrameJob fj)
{
    if (blockIdx.x >= (unsigned)fj.nblocks) {
        stencil9_tile_body<NTS>(in, out, k, ld, x0, x1, y0, y1, c_first, nxw, blockIdx.x - fj.nblocks);
        return;
    }
    auto put = [](double *ptr, double val) { __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const long total = frame_cells(fj.fx1 - fj.fx0 + 1, fj.fy1 - fj.fy0 + 1);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)fj.nblocks * blockDim.x) {
        int i, j;
        frame_index(t, fj.fx0, fj.fx1, fj.fy0, fj.fy1, i, j);
        const size_t o = (size_t)j * ld + i;
        const double r = point9(k, in[o - ld - 1], in[o - ld], in[o - ld + 1], in[o - 1], in[o], in[o + 1],
                                in[o + ld - 1], in[o + ld], in[o + ld + 1]);
        put(out + o, r);
        for (int q = 0; q < fj.pk.n; q++)
            if (i == fj.pk.s[q].i && j >= fj.pk.s[q].j0 && j < fj.pk.s[q].j0 + fj.pk.s[q].nj)
                put(fj.pk.buf + fj.pk.s[q].off + (j - fj.pk.s[q].j0), r);
    }
    __builtin_amdgcn_s_waitcnt(0);        // this wave's stores have been acknowledged ...
    __syncthreads();                      // ... and those of every wave of the group ...
    if (threadIdx.x == 0) {               // ... before the group is counted as done
        const unsigned done = __hip_atomic_fetch_add(fj.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)fj.nblocks - 1) {
            __hip_atomic_store(fj.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fj.flag, fj.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// one cell per thread: odd leading dimensions and unaligned bases
__global__ __launch_bounds__(256) void stencil9_direct(const double *__restrict__ in, double *__restrict__ out, Coef9 k,
                                                      int ld, int x0, int x1, int y0, int y1)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > x1) return;
    for (int j = y0 + blockIdx.y; j <= y1; j += gridDim.y) {
        const size_t o = (size_t)j * ld + i;
        out[o] = point9(k, in[o - ld - 1], in[o - ld], in[o - ld + 1], in[o - 1], in[o], in[o + 1], in[o + ld - 1],
                        in[o + ld], in[o + ld + 1]);
    }
}

// the one-cell frame of the box, one cell per thread, all four sides; west/east column cells also into
// their send-buffer slot (pack loop order, parallel_comms_mod.f90:1678-1683)
__global__ __launch_bounds__(256) void stencil9_frame_k(const double *__restrict__ in, double *__restrict__ out, Coef9 k,
                                                       int ld, int x0, int x1, int y0, int y1, FramePack pk)
{
    const long total = frame_cells(x1 - x0 + 1, y1 - y0 + 1);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int i, j;
        frame_index(t, x0, x1, y0, y1, i, j);
        const size_t o = (size_t)j * ld + i;
        const double r = point9(k, in[o - ld - 1], in[o - ld], in[o - ld + 1], in[o - 1], in[o], in[o + 1],
                                in[o + ld - 1], in[o + ld], in[o + ld + 1]);
        out[o] = r;
        for (int q = 0; q < pk.n; q++)
            if (i == pk.s[q].i && j >= pk.s[q].j0 && j < pk.s[q].j0 + pk.s[q].nj) pk.buf[pk.s[q].off + (j - pk.s[q].j0)] = r;
    }
}

} // namespace

int launch_stencil9_frame(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                          int ystart, int ystop, hipStream_t s, const FramePack *pack)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("stencil9 frame", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && coef != nullptr && in != out, "stencil9: null or aliased arrays");
    const Coef9 k{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8]};
    const long cells = 2L * (xstop - xstart + 1) + 2L * (ystop - ystart + 1);
    int blocks = (int)((cells + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    FramePack pk{};
    if (pack) pk = *pack;
    hipLaunchKernelGGL(stencil9_frame_k, dim3(blocks), dim3(256), 0, s, in, out, k, ld, xstart - 1, xstop - 1, ystart - 1,
                       ystop - 1, pk);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// Frame of the box + interior sweep in ONE launch; *fused = false (nothing launched) when the arrays do not
// qualify for the 16-byte-lane tile kernel or the box has no interior: the caller then takes the two-launch path.
int launch_stencil9_framed(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                           int ystart, int ystop, FrameJob job, hipStream_t s, bool *fused)
{
    *fused = false;
    if (xstop - xstart < 2 || ystop - ystart < 2) return DLESM_OK;
    if (int rc = check_box("dlesm_stencil9_step_dm", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && coef != nullptr && in != out, "stencil9: null or aliased arrays");
    DLESM_REQUIRE(job.counter != nullptr && job.flag != nullptr, "stencil9 framed: no signal words");
    if (!(ld % 2 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && tuning("s9_kernel", 0) == 0)) return DLESM_OK;
    const Coef9 k{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8]};
    job.fx0 = xstart - 1, job.fx1 = xstop - 1, job.fy0 = ystart - 1, job.fy1 = ystop - 1;
    const int x0 = xstart, x1 = xstop - 2, y0 = ystart, y1 = ystop - 2;      // the interior, 0-based
    const int c_first = (x0 / 2) & ~7, c_last = x1 / 2;
    int nxw = (c_last - c_first + 64) / 64, tpb = 4;
    choose_block_shape(&nxw, &tpb);
    const int strips = (y1 - y0 + R) / R;
    const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
    const long cells = 2L * (job.fx1 - job.fx0 + 1) + 2L * (job.fy1 - job.fy0 + 1);
    long nb = ((cells + 64 * tpb - 1) / (64 * tpb) + 7) & ~7L;               // a multiple of 8: tile groups keep their XCD
    job.nblocks = (int)(nb < 8 ? 8 : nb > 256 ? 256 : nb);
    if (nt_stores_for(ld, y0, y1))
        hipLaunchKernelGGL(stencil9_tile_framed<true>, dim3(grid + job.nblocks), dim3(64 * tpb), 0, s, in, out, k, ld, x0, x1,
                           y0, y1, c_first, nxw, job);
    else
        hipLaunchKernelGGL(stencil9_tile_framed<false>, dim3(grid + job.nblocks), dim3(64 * tpb), 0, s, in, out, k, ld, x0, x1,
                           y0, y1, c_first, nxw, job);
    DLESM_HIP_TRY(hipGetLastError());
    *fused = true;
    return DLESM_OK;
}

int launch_stencil9(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                    int ystart, int ystop, hipStream_t s)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK;   // empty box: a zero-trip loop nest
    if (int rc = check_box("dlesm_stencil9_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && coef != nullptr && in != out, "stencil9: null or aliased arrays");
    const Coef9 k{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8]};
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    const bool tile = ld % 2 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && tuning("s9_kernel", 0) == 0;
    if (tile) {
        const int c_first = (x0 / 2) & ~7, c_last = x1 / 2;  // tiles anchored on 128-byte lines of the row
        int nxw = (c_last - c_first + 64) / 64, tpb = 4;
        shape_for_tile_sweep(ld, x0, x1, y0, y1, &nxw, &tpb);
        const int strips = (y1 - y0 + R) / R;
        const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
        if (nt_stores_for(ld, y0, y1))
            hipLaunchKernelGGL(stencil9_tile<true>, dim3(grid), dim3(64 * tpb), 0, s, in, out, k, ld, x0, x1, y0, y1, c_first, nxw);
        else
            hipLaunchKernelGGL(stencil9_tile<false>, dim3(grid), dim3(64 * tpb), 0, s, in, out, k, ld, x0, x1, y0, y1, c_first, nxw);
    } else {
        const int h = y1 - y0 + 1;
        hipLaunchKernelGGL(stencil9_direct, dim3((x1 - x0 + 256) / 256, h > 4096 ? 4096 : h), dim3(256), 0, s, in, out, k,
                           ld, x0, x1, y0, y1);
    }
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

} // namespace dlesm

using namespace dlesm;

extern "C" int dlesm_stencil9_f64(const double *in, double *out, const double *coef, int ld, int ny, int xstart,
                                  int xstop, int ystart, int ystop, void *stream)
{
    if (int rc = ensure_device()) return rc;
    return launch_stencil9(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, (hipStream_t)stream);
}
