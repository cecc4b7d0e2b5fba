// CDNA4 (gfx950) kernels of the dl_esm_inf hot path: the PSy-layer loop nests
// over r2d_field%data, re-done as HBM-streaming HIP kernels.
//
// All of them are memory bound (<= 0.25 flop/byte): no MFMA, no GEMM reshaping.
// What matters is that every cell is fetched from HBM once, in 16-byte-per-lane
// coalesced requests, with enough requests in flight per CU.
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "dlesm_internal.h"
#include "dlesm_device.h"

// 1: tile indices of jacobi5_tile on the scalar unit (readfirstlane of the wave number).  Measured
// A/B on one box at 16384^2: 0.732 ms per launch against 0.718 ms with the per-lane form -- the
// scalar chain delays the first loads of these short-lived waves -- so the default stays 0.  (The
// long fused-step kernel in dlesm_jacobi_x2.hip does gain from it.)
#ifndef DLESM_J5_SCALAR
#define DLESM_J5_SCALAR 0
#endif

namespace dlesm {

// ===========================================================================
// 5-point Jacobi, "march in y" form.
//
// A wave owns 64*VEC consecutive columns and walks up a strip of rows keeping
// the three live rows (jj-1, jj, jj+1) in registers, so each input row is
// loaded once per strip.  West/east neighbours come from the adjacent lane by
// a wave64 shuffle; the two values a wave cannot get from its own lanes (left
// of lane 0, right of lane 63) are fetched by those two lanes with one extra
// 8-byte load per row (an L1/L2 hit: the line is being streamed by the
// neighbouring wave).  Rows are processed U at a time so that U row loads are
// in flight per lane before the first is consumed.
//
// Chunks are anchored at column 0 of the padded row: with ld even (any
// DL_ESM_ALIGNMENT that is a multiple of 2) and a 16-byte aligned base every
// lane's double2 is 16-byte aligned although the interior starts at column 1
// (internal%xstart = 2, parallel_mod.f90:273).  Odd ld falls back to VEC = 1.
// ===========================================================================

template <int VEC> struct Vec;
template <> struct Vec<1> { double v[1]; };
template <> struct __attribute__((aligned(16))) Vec<2> { double v[2]; };

template <int VEC, bool NT>
__device__ __forceinline__ Vec<VEC> load_chunk(const double *p)
{
    Vec<VEC> r;
    if constexpr (VEC == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 t = NT ? __builtin_nontemporal_load((const d2 *)p) : *(const d2 *)p;
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
        r.v[0] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    return r;
}

template <int VEC, bool NT>
__device__ __forceinline__ void store_chunk(double *p, const double (&o)[VEC], bool m0, bool m1)
{
    if constexpr (VEC == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        if (m0 && m1) {
            d2 t = {o[0], o[1]};
            if (NT) __builtin_nontemporal_store(t, (d2 *)p);
            else *(d2 *)p = t;
        } else {
            if (m0) p[0] = o[0];
            if (m1) p[1] = o[1];
        }
    } else {
        if (m0) {
            if (NT) __builtin_nontemporal_store(o[0], p);
            else p[0] = o[0];
        }
    }
}

// one output row from the three live rows; `edge` is meaningful in lanes 0 and 63 only
template <int VEC>
__device__ __forceinline__ void jacobi_row(const Vec<VEC> &south, const Vec<VEC> &mid,
                                           const Vec<VEC> &north, double edge, int lane,
                                           double (&o)[VEC])
{
#ifndef DLESM_J5_SHFL
    // whole-wave shifts on the VALU (DPP) rather than ds_bpermute through the LDS crossbar; lanes 0 / 63 get 0
    // from it and their real value from `edge` just below
    double west = from_lower<true>(mid.v[VEC - 1]);
    double east = from_upper<true>(mid.v[0]);
#else
    double west = __shfl_up(mid.v[VEC - 1], 1);
    double east = __shfl_down(mid.v[0], 1);
#endif
    if (lane == 0) west = edge;
    if (lane == 63) east = edge;
    if constexpr (VEC == 2) {
        o[0] = 0.25 * ((west + mid.v[1]) + (south.v[0] + north.v[0]));
        o[1] = 0.25 * ((mid.v[0] + east) + (south.v[1] + north.v[1]));
    } else {
        o[0] = 0.25 * ((west + east) + (south.v[0] + north.v[0]));
    }
}

#ifdef DLESM_LAB      // the y-march sweep: a comparison point (j5_kernel = 1), libdlesm_hip_lab.so only
// x0..x1, y0..y1: 0-based inclusive interior box.  c_first: first chunk holding an
// interior column.  nxb: blocks per strip.  rows: strip height.
//
// PIPE: register double buffering.  The loads of group g+1 are ISSUED before the
// stores of group g: vmcnt retires loads and stores in issue order, so a wave that
// issues its next loads only after its stores cannot consume them until those stores
// have been acknowledged, and streams in bursts.  flags bit0 (diagnostic builds of a
// profile run only): skip the two edge loads, to price them -- results are then wrong
// at wave boundaries.
template <int VEC, int U, bool NT, bool PIPE>
__global__ __launch_bounds__(256) void jacobi5_march(const double *__restrict__ in,
                                                     double *__restrict__ out, int ld, int x0,
                                                     int x1, int y0, int y1, int c_first, int nxb,
                                                     int rows, int flags)
{
    const int bx = blockIdx.x % nxb, by = blockIdx.x / nxb;
    const int lane = threadIdx.x & 63;
    const int c = c_first + bx * 256 + (int)threadIdx.x; // this lane's chunk
    const int c_last = x1 / VEC;                         // last chunk holding an interior column
    // a wave whose first chunk lies beyond the last interior chunk has nothing to do
    if (c - lane > c_last) return;
    const int c_ld = ld / VEC - 1;                       // last whole chunk inside a row
    const int cl = c < c_ld ? c : c_ld;                  // clamp loads of trailing lanes
    const int col = cl * VEC;
    const bool m0 = c <= c_last && c * VEC >= x0 && c * VEC <= x1;
    const bool m1 = VEC == 2 && c <= c_last && c * VEC + 1 >= x0 && c * VEC + 1 <= x1;
    // which single extra column this lane fetches per row (-1: none)
    int ecol = -1;
    if (lane == 0 && m0) ecol = c * VEC - 1;
    if (lane == 63 && (VEC == 2 ? m1 : m0)) ecol = c * VEC + VEC;
    if (flags & 1) ecol = -1;

    const int jb = y0 + by * rows;
    int je = jb + rows - 1;
    if (je > y1) je = y1;
    int j = jb;

    const double *pin = in + col;
    double *pout = out + (size_t)c * VEC;
    auto row = [&](int jj) { return load_chunk<VEC, NT>(pin + (size_t)jj * ld); };
    auto edge_of = [&](int jj) { return ecol >= 0 ? in[(size_t)jj * ld + ecol] : 0.0; };

    Vec<VEC> south = row(j - 1), mid = row(j);
    double emid = edge_of(j);
    // rows jbase+1 .. jbase+U: the north rows of outputs jbase .. jbase+U-1
    auto load_group = [&](Vec<VEC>(&g)[U], double(&e)[U], int jbase) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            g[u] = row(jbase + 1 + u);
            e[u] = edge_of(jbase + 1 + u);
        }
    };
    auto compute_group = [&](Vec<VEC>(&g)[U], double(&e)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            double o[VEC];
            jacobi_row<VEC>(south, mid, g[u], emid, lane, o);
            store_chunk<VEC, NT>(pout + (size_t)(j + u) * ld, o, m0, m1);
            south = mid;
            mid = g[u];
            emid = e[u];
        }
        j += U;
    };

    int remaining = je - jb + 1;
    if constexpr (PIPE) {
        Vec<VEC> A[U], B[U];
        double eA[U], eB[U];
        if (remaining >= U) {
            load_group(A, eA, j);
            for (;;) {
                bool more = remaining >= 2 * U;
                if (more) load_group(B, eB, j + U);
                compute_group(A, eA);
                remaining -= U;
                if (!more) break;
                more = remaining >= 2 * U;
                if (more) load_group(A, eA, j + U);
                compute_group(B, eB);
                remaining -= U;
                if (!more) break;
            }
        }
    } else {
        while (remaining >= U) {
            Vec<VEC> A[U];
            double eA[U];
            load_group(A, eA, j);
            compute_group(A, eA);
            remaining -= U;
        }
    }
    for (; remaining > 0; remaining--) {
        Vec<VEC> north = row(j + 1);
        double en = edge_of(j + 1);
        double o[VEC];
        jacobi_row<VEC>(south, mid, north, emid, lane, o);
        store_chunk<VEC, NT>(pout + (size_t)j * ld, o, m0, m1);
        south = mid;
        mid = north;
        emid = en;
        j++;
    }
}

#endif // DLESM_LAB

// ===========================================================================
// 5-point Jacobi, linear tile sweep (the default).
//
// Measured on MI355X (scripts/membench.hip): a read+write stream reaches its best
// rate, 6.3 TB/s, when the workgroups of a launch sweep memory linearly in
// dispatch order with short-lived groups; long-lived groups marching far apart
// lose 15 %.  So wave tiles of 64*VEC columns x R rows are numbered row-major over
// the box and a workgroup is blockDim/64 consecutive tiles:
//   * the chip reads one narrow band of rows and writes another, front to back;
//   * all R+2 row loads of a tile are issued before the first is consumed
//     (straight-line code, no loop-carried vmcnt wait), (R+2) KiB in flight/wave;
//   * waves are independent (no LDS, no barrier), so a ragged last column costs one
//     partial wave per strip;
//   * the rows a tile shares with the tile below are re-read from L2 when the two
//     run on the same XCD (workgroups are dealt round-robin to the 8 XCDs) and from
//     the Infinity Cache otherwise -- choose_block_shape() picks the shape.
// Per-XCD row bands and an XCD-affine column-major tile order were built, measured
// (70 %, 69-70 % against 75-78 %; profiles/r01_sweep_tile_group.txt) and removed.
// ===========================================================================
// NT: cache policy -- bit 0: non-temporal loads, bit 1: non-temporal stores.  Loads never pay (the rows are
// re-read by the tile below); stores do once the two arrays no longer fit the Infinity Cache -- see launch_tile.
template <int VEC, int R, int NT>
__device__ __forceinline__ void jacobi5_tile_body(const double *__restrict__ in, double *__restrict__ out,
                                                  int ld, int x0, int x1, int y0, int y1, int c_first,
                                                  int nxw, int flags, unsigned block)
{
    const int lane = threadIdx.x & 63;
#if DLESM_J5_SCALAR
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const int wv = threadIdx.x >> 6;
#endif
    const int w = block * (blockDim.x >> 6) + wv;       // wave-tile number
    int xw = w % nxw, jb = y0 + (w / nxw) * R;
    const int by1 = y1;
#if DLESM_J5_SCALAR
    xw = __builtin_amdgcn_readfirstlane(xw);           // (integer division runs on the VALU)
    jb = __builtin_amdgcn_readfirstlane(jb);
#endif
    if (jb > by1) return;
    int je = jb + R - 1;
    if (je > by1) je = by1;

    const int c = c_first + xw * 64 + lane;
    if (c - lane > x1 / VEC) return;                    // an idle padding tile
    const int c_last = x1 / VEC;
    const int c_ld = ld / VEC - 1;
    const int cl = c < c_ld ? c : c_ld;
    const bool m0 = c <= c_last && c * VEC >= x0 && c * VEC <= x1;
    const bool m1 = VEC == 2 && c <= c_last && c * VEC + 1 >= x0 && c * VEC + 1 <= x1;
    int ecol = -1;
    if (lane == 0 && m0) ecol = c * VEC - 1;
    if (lane == 63 && (VEC == 2 ? m1 : m0)) ecol = c * VEC + VEC;
    if (flags & 1) ecol = -1;

    const double *pin = in + (size_t)cl * VEC;
    double *pout = out + (size_t)c * VEC;
    // rows jb-1 .. jb+R ; rows beyond je+1 (short last strip of a band) are clamped:
    // they are loaded again but never used
    Vec<VEC> r[R + 2];
    double e[R + 2];
#pragma unroll
    for (int u = 0; u < R + 2; u++) {
        int jj = jb - 1 + u;
        if (jj > je + 1) jj = je + 1;
        r[u] = load_chunk<VEC, (NT & 1) != 0>(pin + (size_t)jj * ld);
        e[u] = ecol >= 0 ? in[(size_t)jj * ld + ecol] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
        if (jb + u <= je) {
            double o[VEC];
            jacobi_row<VEC>(r[u], r[u + 1], r[u + 2], e[u + 1], lane, o);
            store_chunk<VEC, (NT & 2) != 0>(pout + (size_t)(jb + u) * ld, o, m0, m1);
        }
    }
}

template <int VEC, int R, int NT>
__global__ __launch_bounds__(1024) void jacobi5_tile(const double *__restrict__ in,
                                                    double *__restrict__ out, int ld, int x0, int x1,
                                                    int y0, int y1, int c_first, int nxw, int flags)
{
    jacobi5_tile_body<VEC, R, NT>(in, out, ld, x0, x1, y0, y1, c_first, nxw, flags, blockIdx.x);
}

// WT: store with device-scope write-through (relaxed agent-scope atomic stores), for frame cells that
// another kernel reads while this one is still running.
template <bool WT>
__device__ __forceinline__ void frame_cell(long t, const double *__restrict__ in, double *__restrict__ out,
                                           int ld, int x0, int x1, int y0, int y1, const FramePack &pk,
                                           const FrameJob *fj = nullptr)
{
    int i, j;
    frame_index(t, x0, x1, y0, y1, i, j);
    const size_t o = (size_t)j * ld + i;
    // WT (frame inside the interior launch): the operand OUTSIDE the box is a halo cell the exchange may
    // have written after this kernel started -- read it at device scope, past this XCD's L2.  The
    // operands inside the box were written by the previous launch: ordinary cached loads.
    auto get = [](const double *p, bool outside) {
        if constexpr (WT) {
            if (outside) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return *p;
    };
    // a west/east halo operand that still sits in the receive buffer (pipelined steps: no unpack)
    auto get_col = [&](int hi, const double *p, bool outside) {
        if constexpr (WT) {
            if (outside && fj && fj->halo_buf)
                for (int k = 0; k < fj->nh; k++)
                    if (hi == fj->hs[k].i && j >= fj->hs[k].j0 && j < fj->hs[k].j0 + fj->hs[k].nj)
                        return __hip_atomic_load(fj->halo_buf + fj->hs[k].off + (j - fj->hs[k].j0), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
        return get(p, outside);
    };
    const double r = 0.25 * ((get_col(i - 1, in + o - 1, i == x0) + get_col(i + 1, in + o + 1, i == x1)) +
                             (get(in + o - ld, j == y0) + get(in + o + ld, j == y1)));
    auto put = [](double *p, double v) {
        if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
    put(out + o, r);
    for (int k = 0; k < pk.n; k++)
        if (i == pk.s[k].i && j >= pk.s[k].j0 && j < pk.s[k].j0 + pk.s[k].nj)
            put(pk.buf + pk.s[k].off + (j - pk.s[k].j0), r);
}

// The distributed step in ONE launch on the caller's stream: the first fj.nblocks workgroups
// compute the frame of the box (fx0:fx1, fy0:fy1) -- the cells the neighbours are waiting for --
// and publish `seq` in a device-memory flag when the last of them is done (the library's side
// stream holds a one-wave kernel that sleeps on that flag, frame_flag_wait, and starts the exchange
// then); all other workgroups are the ordinary linear tile sweep over the interior (x0:x1, y0:y1).
// No frame launch and no event record between frame and interior on the caller's stream: measured,
// those cost it ~5 us of a 180 us step (scripts/syncbench.hip, profiles/r02_syncbench.txt).
template <int VEC, int R, int NT>
__global__ __launch_bounds__(1024) void jacobi5_tile_framed(const double *__restrict__ in,
                                                           double *__restrict__ out, int ld, int x0, int x1,
                                                           int y0, int y1, int c_first, int nxw, int flags,
                                                           FrameJob fj)
{
    if (blockIdx.x >= (unsigned)fj.nblocks) {
        jacobi5_tile_body<VEC, R, NT>(in, out, ld, x0, x1, y0, y1, c_first, nxw, flags, blockIdx.x - fj.nblocks);
        return;
    }
    const long total = frame_cells(fj.fx1 - fj.fx0 + 1, fj.fy1 - fj.fy0 + 1);
    if (fj.halo_seq) {
        // pipelined steps: `in`'s halos (and the send buffer) belong to the previous step's exchange
        // until its completion flag is up.  In steady state it has been up for a long time -- the
        // exchange ends well inside the previous interior sweep -- and this costs one load.  The flag depends on the
        // OTHER ranks: bounded by dm_wait_seconds (default 10 min, 0 = no limit), not by the 30 s of a local wait.
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(fj.halo_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fj.halo_seq) {
                __builtin_amdgcn_s_sleep(32);
                if (fj.halo_wait_ticks && __builtin_amdgcn_s_memrealtime() - t0 > fj.halo_wait_ticks) {   // 100 MHz counter
                    __hip_atomic_store(fj.timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            // The halos were written by kernels that have ENDED (RCCL's receive, the unpack: plain stores, released at
            // device scope when they ended, the flag stored by a later kernel), but this workgroup was already running:
            // one acquire, waited for, then the barrier -- the guide's consumer form.  What it invalidates is this CU's
            // L1 (and what this XCD's L2 may hold of lines other XCDs have rewritten); it costs ~2 us once per frame
            // workgroup, beside the sweep (dm_acquire = 0 takes it out, for measurements only).
            if (fj.acquire) handover_acquire<false>();
        }
        __syncthreads();
    }
    // The frame cells are read by the exchange while this kernel is still running, possibly from
    // another XCD (whose L2 is not coherent with this one): they are stored write-through at device
    // scope.  A per-thread __threadfence() instead would be a whole-L2 write-back + invalidate per
    // wave, in the middle of the interior sweep -- measured: +35 us on a 180 us step.
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)fj.nblocks * blockDim.x)
        frame_cell<true>(t, in, out, ld, fj.fx0, fj.fx1, fj.fy0, fj.fy1, fj.pk, &fj);
    __builtin_amdgcn_s_waitcnt(0);        // this wave's stores have been acknowledged ...
    __syncthreads();                      // ... and those of every wave of the group ...
    if (threadIdx.x == 0) {               // ... before the group is counted as done
        const unsigned done = __hip_atomic_fetch_add(fj.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)fj.nblocks - 1) {
            // ready for the next launch (launches on one plan are stream ordered)
            __hip_atomic_store(fj.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fj.flag, fj.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- peer transport: the frame workgroups ARE the exchange (PeerJob, dlesm_internal.h) ----------------------------
// Payload and flags cross GPUs: every access to a mailbox is a relaxed SYSTEM-scope atomic (`global_load/store ... sc0
// sc1`: nothing of it stays in an L1 or L2 on either side), the mailboxes are fine-grained allocations, and the order
// payload -> flag is the one of jacobi5_tile_framed: every storing wave drains (`s_waitcnt vmcnt(0)`: the stores have
// been acknowledged by the memory they went to), the workgroup's barrier, one device-scope counter increment per
// workgroup, and the LAST arriver stores the flags.
// 16 bytes into a mailbox with ONE system-scope store (`global_store_dwordx4 ... sc0 sc1`): over xGMI every store is its own
// packet, and the guide prices an 8-byte sc1 store at 2.7x a 16-byte one per byte.  A volatile access through an
// address-space-1 pointer is what the compiler lowers to exactly that instruction (checked in the ISA; a plain volatile
// pointer gives flat_store); dst must be 16-byte aligned.
typedef double pd2 __attribute__((ext_vector_type(2)));
typedef double pd2a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void peer_store16(double *dst, double a, double b)
{
    *(__attribute__((address_space(1))) volatile pd2 *)dst = pd2{a, b};
}
__device__ __forceinline__ void peer_store8(double *dst, double a)
{
    __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one frame cell of the distributed Jacobi step: halo operands from the local mailbox while the previous step's strips
// have not been unpacked (pj.virt), everything else from the field
__device__ __forceinline__ double peer_cell_value(int i, int j, const double *__restrict__ in, int ld, const PeerJob &pj)
{
    auto operand = [&](int oi, int oj, bool outside) {
        if (outside && pj.virt)
            for (int k = 0; k < pj.nin; k++) {
                const PeerJob::In &m = pj.in[k];
                if (oi >= m.i0 && oi < m.i0 + m.ni && oj >= m.j0 && oj < m.j0 + m.nj)
                    return __hip_atomic_load(m.src + (size_t)(oj - m.j0) * m.ni + (oi - m.i0), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
            }
        return in[(size_t)oj * ld + oi];          // inside the box, a fixed boundary cell, or halos already unpacked
    };
    return 0.25 * ((operand(i - 1, j, i == pj.fx0) + operand(i + 1, j, i == pj.fx1)) +
                   (operand(i, j - 1, j == pj.fy0) + operand(i, j + 1, j == pj.fy1)));
}

// The frame workgroups of the peer step (round 4: WIDE stores into the neighbours' mailboxes).  Work items, numbered so
// that the 64 items of a wave always lie in ONE section (sections are padded to the workgroup size):
//   rows     south, north: an item = TWO adjacent cells of the row (i, i+1): one 16-byte store into out, and for every
//            row strip that holds both at an even offset ONE 16-byte system-scope store into the neighbour's mailbox (a
//            strip's slot starts on a 128-byte line); ragged ends and 1 x 1 corner strips take 8-byte stores;
//   columns  west, east, over the FULL height (the two corner cells are computed again -- same operands, same bits -- so
//            that a column strip is written by its column's lanes alone): an item = one cell, gathered at stride ld.  A
//            column strip is contiguous in the mailbox (j order = the pack loop's order, parallel_comms_mod.f90:1678-1683),
//            so the lane of an EVEN slot takes its upper neighbour's value by a wave shift (DPP, no LDS, no barrier) and
//            stores both: a wave covers 512 bytes of the neighbour's memory with 32 stores instead of 64.
// Per step and rank that halves the stores that cross xGMI: (w + h) 16-byte stores instead of 2 (w + h) 8-byte ones.
// (A form that staged each column chunk in LDS and let half the lanes write it -- two barriers per turn, 8 KB of LDS on
// every workgroup of the launch -- gave the same bits and cost the loop-back step 1 %; this one costs nothing.)
__device__ __forceinline__ void peer_frame_wide(const double *__restrict__ in, double *__restrict__ out, int ld,
                                                const PeerJob &pj, unsigned block)
{
    const int w = pj.fx1 - pj.fx0 + 1, h = pj.fy1 - pj.fy0 + 1, B = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    const int nrows = h > 1 ? 2 : 1, ncols = w > 1 ? 2 : 1;
    const long rp = ((long)(w + 1) / 2 + B - 1) / B * B;         // pair items per row, padded to whole workgroup turns
    const long cp = ((long)h + B - 1) / B * B;                   // cell items per column, likewise
    const long row_items = nrows * rp, total = row_items + ncols * cp;
    for (long base = (long)block * B; base < total; base += (long)pj.nblocks * B) {      // (uniform per workgroup)
        if (base < row_items) {
            const long t = base + tid;
            const int r = (int)(t / rp), i = pj.fx0 + 2 * (int)(t - (long)r * rp), j = r == 0 ? pj.fy0 : pj.fy1;
            const bool v0 = i <= pj.fx1, v1 = i + 1 <= pj.fx1;
            if (!v0) continue;
            const double a = peer_cell_value(i, j, in, ld, pj), b = v1 ? peer_cell_value(i + 1, j, in, ld, pj) : 0.0;
            double *po = out + (size_t)j * ld + i;                // read by the next launch: ordinary stores
            if (v1) *(pd2a8 *)po = pd2{a, b};
            else po[0] = a;
            for (int k = 0; k < pj.nout; k++) {
                const PeerJob::Out &m = pj.out[k];
                if (m.nj != 1 || j != m.j0) continue;             // (column strips: the column items below)
                const bool in0 = i >= m.i0 && i < m.i0 + m.ni, in1 = v1 && i + 1 >= m.i0 && i + 1 < m.i0 + m.ni;
                double *d = m.dst + (i - m.i0);
                if (in0 && in1 && ((i - m.i0) & 1) == 0) peer_store16(d, a, b);
                else {
                    if (in0) peer_store8(d, a);
                    if (in1) peer_store8(d + 1, b);
                }
            }
        } else {
            const long t = base - row_items + tid;
            const int side = (int)(t / cp), i = side == 0 ? pj.fx0 : pj.fx1, j = pj.fy0 + (int)(t - (long)side * cp);
            const bool valid = j <= pj.fy1;
            const double v = valid ? peer_cell_value(i, j, in, ld, pj) : 0.0;
            if (valid && j > pj.fy0 && j < pj.fy1) out[(size_t)j * ld + i] = v;   // (the corner cells are stored by the rows)
            const double up = from_upper<true>(v);                // lane + 1's cell = row j + 1 (every lane of the wave is here)
            for (int k = 0; k < pj.nout; k++) {
                const PeerJob::Out &m = pj.out[k];
                if (m.ni != 1 || m.nj == 1 || m.i0 != i) continue;
                const int sl = j - m.j0;                         // this cell's slot in the strip
                if (!valid || sl < 0 || sl >= m.nj) continue;
                if ((sl & 1) == 0) {                             // even slot: 16 bytes with the upper neighbour, where there is one
                    if (lane < 63 && j + 1 <= pj.fy1 && sl + 1 < m.nj) peer_store16(m.dst + sl, v, up);
                    else peer_store8(m.dst + sl, v);
                } else if (lane == 0 || sl == 0) {               // odd slot whose lower neighbour could not take it along
                    peer_store8(m.dst + sl, v);
                }
            }
        }
    }
}

// The arrival flags of an operation's messages, raised by the ONE lane that has seen every storing workgroup report (each
// behind its waves' `s_waitcnt vmcnt(0)` and its barrier).  fenced (tuning mailbox_fences, default 1): the guide's producer
// form at system scope -- release fence, an asm wait the compiler cannot drop, then the relaxed flag stores -- and on the
// other side one acquire behind the poll (peer_wait_flags): flag -> payload is then a release / acquire pair of the memory
// model whatever the page attributes of the mailbox.  fenced = 0 (measurements; the form the round-3 loop-back figures were
// taken with): the flags follow payload stores that are write-through system-scope stores into uncached memory, drained --
// enough by what the hardware was seen to do on ONE GPU, not a guarantee, and never exercised across xGMI.
__device__ __forceinline__ void peer_publish(bool fenced)
{
    if (fenced) handover_release<true>();
}
__device__ __forceinline__ void peer_raise_flag(unsigned long long *flag, unsigned long long seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait of ONE thread for n arrival flags (system scope: they are stored by other GPUs); the caller's workgroup
// passes __syncthreads() before any of its lanes loads a strip
__device__ __forceinline__ void peer_wait_flags(const PeerJob::In *in, int n, unsigned long long seq,
                                                unsigned long long ticks, int *timed_out, bool fenced)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; k++)
        while (__hip_atomic_load(in[k].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(32);
            if (ticks && __builtin_amdgcn_s_memrealtime() - t0 > ticks) {                   // 100 MHz counter
                __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
    if (fenced) handover_acquire<true>();      // see peer_publish
}

// SGPR budget: a 256-thread workgroup is admitted 8 per CU only up to 80 SGPRs (82-96: 7; MI355X_MICROARCH.md, "Residency"),
// and the occupancy of the whole launch -- the tile sweep included -- follows the kernel's count.  The frame code's message
// tables took it to 86 and the 8192^2 loop-back step lost 0.9 %; capped, the compiler keeps the excess in VGPR lanes.
template <int VEC, int R, int NT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(72))) void jacobi5_tile_peer(const double *__restrict__ in, double *__restrict__ out, int ld,
                                                         int x0, int x1, int y0, int y1, int c_first, int nxw, int flags,
                                                         PeerJob pj)
{
    if (blockIdx.x >= (unsigned)(pj.nblocks + pj.nunb)) {
        jacobi5_tile_body<VEC, R, NT>(in, out, ld, x0, x1, y0, y1, c_first, nxw, flags, blockIdx.x - pj.nblocks - pj.nunb);
        return;
    }
    if (blockIdx.x >= (unsigned)pj.nblocks) {      // the join inside the launch: wait for this step's strips, copy them into out's halos
        const int b = blockIdx.x - pj.nblocks;
        for (int k = b % pj.nun, part = b / pj.nun; k < pj.nun; k += pj.nunb) {      // (nunb >= nun: one strip per group, several groups per strip)
            const PeerJob::In m = pj.un[k];
            const int parts = (pj.nunb - k + pj.nun - 1) / pj.nun;                  // groups that share strip k
            if (threadIdx.x == 0) peer_wait_flags(&m, 1, peer_seq_load(pj.seqw, pj.seq), pj.wait_ticks, pj.timed_out, pj.fenced != 0);
            __syncthreads();
            const long n = (long)m.ni * m.nj;
            for (long t = (long)part * blockDim.x + threadIdx.x; t < n; t += (long)parts * blockDim.x) {
                const int jj = (int)(t / m.ni), ii = (int)(t - (long)jj * m.ni);
                out[(size_t)(m.j0 + jj) * ld + (m.i0 + ii)] = __hip_atomic_load(m.src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (pj.wait_seq) {      // the neighbours' frames of the previous step: in steady state long since there (one load each)
        if (threadIdx.x == 0) peer_wait_flags(pj.in, pj.nin, peer_seq_load(pj.seqw, pj.seq) - 1, pj.wait_ticks, pj.timed_out, pj.fenced != 0);   // (wait_seq == seq - 1)
        __syncthreads();
    }
    peer_frame_wide(in, out, ld, pj, blockIdx.x);
    __builtin_amdgcn_s_waitcnt(0);        // this wave's stores have been acknowledged ...
    __syncthreads();                      // ... and those of every wave of the group ...
    if (threadIdx.x == 0) {               // ... before the group is counted as done
        const unsigned done = __hip_atomic_fetch_add(pj.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)pj.nblocks - 1) {
            __hip_atomic_store(pj.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long seq = peer_seq_load(pj.seqw, pj.seq);
            peer_publish(pj.fenced != 0);
            for (int k = 0; k < pj.nout; k++)
                peer_raise_flag(pj.out[k].flag, seq);
            peer_seq_advance(pj.seqw, seq, pj.timed_out);
        }
    }
}

// Joins of the peer transport: wait for the strips' arrival flags, then copy them from the mailbox into the halo cells
// of the field (ordinary stores: the readers are later launches on this stream).  grid = (parts, strips).
struct PeerFields { double *f[16]; };
__global__ __launch_bounds__(256) void peer_unpack_k(PeerStrips st, unsigned long long seq, const unsigned long long *seqw, PeerFields fields,
                                                     int ld, unsigned long long ticks, int *timed_out, int fenced)
{
    const PeerJob::In m = st.s[blockIdx.y];
    if (threadIdx.x == 0) peer_wait_flags(&m, 1, peer_seq_load(seqw, seq), ticks, timed_out, fenced != 0);
    __syncthreads();
    const long n = (long)m.ni * m.nj;
    double *__restrict__ field = fields.f[blockIdx.z];
    const double *src = m.src + (long)blockIdx.z * n;            // field after field inside a message
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int jj = (int)(t / m.ni), ii = (int)(t - (long)jj * m.ni);
        field[(size_t)(m.j0 + jj) * ld + (m.i0 + ii)] = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// First half of an exchange over the mailboxes: grid = (parts, strips, fields).  Payload stores and flags as in
// jacobi5_tile_peer: system-scope write-through stores, drained, barrier, one counter increment per workgroup, the last
// arriver raises every neighbour's flag.
__global__ __launch_bounds__(256) void peer_pack_k(PeerOuts out, PeerFields fields, int ld, unsigned *counter,
                                                   unsigned long long seq, unsigned long long *seqw, int *sticky, int fenced)
{
    const PeerJob::Out m = out.s[blockIdx.y];
    const long n = (long)m.ni * m.nj;
    const double *__restrict__ field = fields.f[blockIdx.z];
    double *dst = m.dst + (long)blockIdx.z * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int jj = (int)(t / m.ni), ii = (int)(t - (long)jj * m.ni);
        __hip_atomic_store(dst + t, field[(size_t)(m.j0 + jj) * ld + (m.i0 + ii)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == total - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long dseq = peer_seq_load(seqw, seq);
            peer_publish(fenced != 0);
            for (int k = 0; k < out.n; k++)
                peer_raise_flag(out.s[k].flag, dseq);
            peer_seq_advance(seqw, dseq, sticky);
        }
    }
}

// Both halves of an exchange in ONE launch: grid = (parts, strips, 2 x fields); the first `nf` z-slices are the pack half
// (dispatched first), the other nf the unpack half -- which waits for the NEIGHBOURS' flags and so does not depend on this
// launch's own pack half (nor theirs on ours): no cycle.  One launch less per exchange (7 -> 5 us from a compiled host).
__global__ __launch_bounds__(256) void peer_exchange_k(PeerOuts out, PeerStrips in, PeerFields fields, int nf, int ld,
                                                       unsigned *counter, unsigned long long seq, unsigned long long *seqw,
                                                       unsigned long long ticks, int *timed_out, int fenced)
{
    const int k = blockIdx.z % nf;
    if ((int)blockIdx.z < nf) {                          // ---- pack half
        if ((int)blockIdx.y < out.n) {
            const PeerJob::Out m = out.s[blockIdx.y];
            const long n = (long)m.ni * m.nj;
            const double *__restrict__ field = fields.f[k];
            double *dst = m.dst + (long)k * n;
            for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
                const int jj = (int)(t / m.ni), ii = (int)(t - (long)jj * m.ni);
                __hip_atomic_store(dst + t, field[(size_t)(m.j0 + jj) * ld + (m.i0 + ii)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = gridDim.x * gridDim.y * nf;       // every block of the pack half reports, idle ones too
            const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == total - 1) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long dseq = peer_seq_load(seqw, seq);
                peer_publish(fenced != 0);
                for (int q = 0; q < out.n; q++)
                    peer_raise_flag(out.s[q].flag, dseq);
                peer_seq_advance(seqw, dseq, timed_out);
            }
        }
        return;
    }
    if ((int)blockIdx.y >= in.n) return;                 // ---- unpack half
    const PeerJob::In m = in.s[blockIdx.y];
    if (threadIdx.x == 0) peer_wait_flags(&m, 1, peer_seq_load(seqw, seq), ticks, timed_out, fenced != 0);
    __syncthreads();
    const long n = (long)m.ni * m.nj;
    double *__restrict__ field = fields.f[k];
    const double *src = m.src + (long)k * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int jj = (int)(t / m.ni), ii = (int)(t - (long)jj * m.ni);
        field[(size_t)(m.j0 + jj) * ld + (m.i0 + ii)] = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct PeerFlagList { unsigned long long *f[PeerJob::MAXM]; };
__global__ void peer_flags_set_k(PeerFlagList fl, int n, unsigned long long seq, unsigned long long *seqw, int *sticky, int fenced)
{
    const unsigned long long dseq = peer_seq_load(seqw, seq);
    peer_publish(fenced != 0);            // (behind a kernel boundary the payload is already released; kept for one form everywhere)
    if (threadIdx.x < (unsigned)n) peer_raise_flag(fl.f[threadIdx.x], dseq);
    __syncthreads();                      // (one workgroup: every lane has read the word before lane 0 moves the other one)
    if (threadIdx.x == 0) peer_seq_advance(seqw, dseq, sticky);
}

#ifdef DLESM_LAB
// LDS-staged form (j5_kernel = 2, the comparison point for "stage the tile and its halo ring in
// LDS"): a workgroup of 64*T lanes stages rows jb-1..je+1 of a 128*T-column tile, plus the two
// ring columns, in LDS (row pitch 2*blockDim+4 doubles, interior chunks 16-byte aligned at
// index 2), then every lane reads its five operands from LDS.  Same expression tree, same
// results; what it buys over the register form is a taller tile (R up to 16: (R+2)/R re-read)
// at the price of a barrier and of LDS round trips.  Measured: see DESIGN.md section 5.1.
template <int R>
__global__ __launch_bounds__(1024) void jacobi5_lds(const double *__restrict__ in,
                                                   double *__restrict__ out, int ld, int x0, int x1,
                                                   int y0, int y1, int c_first, int nxb)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double tile[];
    const int tid = threadIdx.x, nt = blockDim.x, pitch = 2 * nt + 4;
    const int xb = blockIdx.x % nxb, jb = y0 + (blockIdx.x / nxb) * R;
    if (jb > y1) return;                                 // block-uniform exits only
    int je = jb + R - 1;
    if (je > y1) je = y1;
    const int c = c_first + xb * nt + tid, c_last = x1 / 2, c_ld = ld / 2 - 1;
    if (c - tid > c_last) return;                        // idle padding block
    const int cl = c < c_ld ? c : c_ld;
    const bool m0 = c <= c_last && c * 2 >= x0 && c * 2 <= x1;
    const bool m1 = c <= c_last && c * 2 + 1 >= x0 && c * 2 + 1 <= x1;
    int ecol = -1, eidx = 0;
    if (tid == 0 && m0) { ecol = c * 2 - 1; eidx = 1; }
    if (tid == nt - 1 && m1) { ecol = c * 2 + 2; eidx = 2 * nt + 2; }
    const double *pin = in + (size_t)cl * 2;
#pragma unroll
    for (int u = 0; u < R + 2; u++) {
        int jj = jb - 1 + u;
        if (jj > je + 1) jj = je + 1;
        *(d2 *)(tile + u * pitch + 2 + 2 * tid) = *(const d2 *)(pin + (size_t)jj * ld);
        if (ecol >= 0) tile[u * pitch + eidx] = in[(size_t)jj * ld + ecol];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < R; u++) {
        if (jb + u <= je) {
            const double *row = tile + (u + 1) * pitch + 2 + 2 * tid;
            const d2 mid = *(const d2 *)row, south = *(const d2 *)(row - pitch), north = *(const d2 *)(row + pitch);
            const double west = row[-1], east = row[2];
            const double o0 = 0.25 * ((west + mid.y) + (south.x + north.x));
            const double o1 = 0.25 * ((mid.x + east) + (south.y + north.y));
            double *po = out + (size_t)(jb + u) * ld + (size_t)c * 2;
            if (m0 && m1) *(d2 *)po = d2{o0, o1};
            else {
                if (m0) po[0] = o0;
                if (m1) po[1] = o1;
            }
        }
    }
}

#endif // DLESM_LAB

// Block shape of a linear tile sweep: waves per workgroup and (padded) wave tiles per row.
// Workgroups go round-robin to the 8 XCDs, so the tile below a given tile runs on the XCD
// (groups per row) mod 8 further on; the rule below is fitted to measurements, see inside.
void choose_block_shape(int *nxw_io, int *tpb_out, int prefer)
{
    int nxw = *nxw_io, tpb = 4, pad = 0;
    int forced = tuning("j5_tpb", 0);
    if (!forced && tuning("j5_autoshape", 1)) forced = prefer;   // the caller's measured block size
    if (!tuning("j5_autoshape", 1)) {                 // experiments: plain shape, no padding
        *tpb_out = (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) ? forced : 4;
        *nxw_io = nxw + tuning("j5_pad_tiles", 0);
        return;
    }
    // Exhaustive searches over (waves per group, tiles per row) at 1024^2 .. 16384^2
    // (scripts/shape_search.py, profiles/r01_shape_search.txt) all show the same thing: the sweep is
    // fast when a row is A QUARTER OF A WORKGROUP short of, or past, a multiple of 8 workgroups
    // (7.75, 8.25, 15.75, 16.25, ... groups per row): the tile below a tile then runs on the same XCD
    // for 3 tiles in 4 (its re-read rows are L2 hits) while the columns an XCD works on still drift
    // from row to row.  At the exact multiple it is up to 40 % slower (presumably every XCD then keeps
    // working on the same address stripes), at +-1/8 group it is sometimes fast and sometimes 14 % slower, in between
    // it is slow.  Idle padding tiles leave at once and cost next to nothing (58 % of them at 10000^2
    // still wins); 8 and 4 waves per group are equally good at the same drift, 2 is worse.
    // So: the smallest tile count >= nxw that is = +-(waves per group)/4 modulo 8 groups.
    if (nxw < 8 && !forced) {                         // thin boxes (frames): plain
        *nxw_io = nxw + tuning("j5_pad_tiles", 0);
        *tpb_out = 4;
        return;
    }
    const bool skew = tuning("j5_skew", 1);
    const double padw = tuning("j5_padw", 2) / 100.0;
    double best = 1e9;
    for (int cand : {8, 4, 2, 16}) {
        if (forced && cand != forced) continue;
        const int period = 8 * cand, r = nxw % period, base = nxw - r, d = cand >= 4 ? cand / 4 : 1;
        int target;
        if (!skew) target = r ? base + period : nxw;                  // experiments: the exact multiple
        else target = r <= d ? base + d : (r <= period - d ? base + period - d : base + period + d);
        const int p = target - nxw;
        const double cost = padw * p / nxw + (cand == 8 ? 0.0 : cand == 4 ? 0.01 : cand == 2 ? 0.05 : 0.04);
        if (cost < best) { best = cost; tpb = cand; pad = p; }
    }
    nxw += pad + tuning("j5_pad_tiles", 0);
    *nxw_io = nxw;
    *tpb_out = tpb;
}

// Measured shapes (dlesm_stencil5_autotune_f64): every shape computes the same bits, so the best
// (waves per group, tiles per row) for a given (leading dimension, box) can simply be timed once
// and remembered -- the fine structure of the landscape depends on the row pitch and is not
// captured by the rule above to better than +-2 %.
struct ShapeKey {
    int ld, x0, x1, y0, y1, vec;
    bool operator<(const ShapeKey &o) const
    {
        return std::tie(ld, x0, x1, y0, y1, vec) < std::tie(o.ld, o.x0, o.x1, o.y0, o.y1, o.vec);
    }
};
struct Shape { int tpb, nxw, rows; };                   // rows = 0: the default tile height
static std::mutex g_shape_mu;
static std::map<ShapeKey, Shape> g_shape_cache;
static Shape g_shape_override = {0, 0, 0};              // set only while the autotuner is measuring

// Store policy of the two-to-three-stream sweeps (Jacobi, 3x3, masked, whole-field copy): non-temporal once
// an array of the box's height no longer shares the 256 MB Infinity Cache with its partner -- from 150 MB
// (measured: 134 MB arrays lose 3-4 % with non-temporal stores, 164 MB gain 0.6 %, 193-376 MB 2.5-4 %, see
// launch_tile and DESIGN.md section 5.1); j5_nt_stores = 1 / 0 forces it on / off.
int nt_stores_for(int ld, int y0, int y1)
{
    const int t = tuning("j5_nt_stores", -1);
    return t >= 0 ? (t != 0) : (size_t)ld * (size_t)(y1 - y0 + 3) * sizeof(double) >= ((size_t)150 << 20);
}

template <int VEC, bool NT>
static void launch_tile(const double *in, double *out, int ld, int x0, int x1, int y0, int y1, int R,
                        int flags, hipStream_t s, FrameJob *fj = nullptr, PeerJob *pj = nullptr)
{
#ifdef DLESM_LAB
    if (R != 1 && R != 2 && R != 3 && R != 4 && R != 6 && R != 12 && R != 16) R = 8;
#else
    // the product's tile heights: 2 or 3 rows with 16-byte lanes (the planning call chooses), 4 rows with the 8-byte-lane
    // fall-back; the other heights were comparison points and live in libdlesm_hip_lab.so
    if (VEC == 2) { if (R != 3) R = 2; }
    else R = 4;
#endif
    // tiles are anchored on a 128-byte line of the row (not on the first interior column), so
    // every wave access covers whole lines whatever the box: lanes left of x0 are masked
    const int c_first = (x0 / VEC) & ~(128 / (8 * VEC) - 1), c_last = x1 / VEC;
    int nxw = (c_last - c_first + 64) / 64;             // wave tiles per row
    int tpb = 4;                                         // tiles (waves) per block
    // Blocks go round-robin to the 8 XCDs, so the tile below a given tile runs on the XCD
    // (blocks per row) mod 8 further on: the re-read of the shared rows is an L2 hit only when
    // that is ~0.  Measured (scripts/pad_probe.py): 32.25 blocks per row is the sweet spot at
    // 16384^2, 33 blocks per row costs 19 %.  Pick the block size (2..16 waves) that brings
    // blocks-per-row closest above a multiple of 8, and skew an exact multiple by one idle tile.
    {
        std::lock_guard<std::mutex> lk(g_shape_mu);
        // (the framed launch is built for 2-row tiles: it takes the best 2-row shape, kept under VEC + 100)
        auto it = g_shape_cache.find(ShapeKey{ld, x0, x1, y0, y1, (fj || pj) ? VEC + 100 : VEC});
        int rows = 0;
        if (g_shape_override.tpb) { tpb = g_shape_override.tpb; nxw = g_shape_override.nxw; rows = g_shape_override.rows; }
        else if (it != g_shape_cache.end() && tuning("j5_use_tuned", 1)) { tpb = it->second.tpb; nxw = it->second.nxw; rows = it->second.rows; }
        else choose_block_shape(&nxw, &tpb);
        // a measured tile height (2 or 3 rows: which streams better depends on the row pitch) -- not for the
        // framed launch, whose kernel is built for 2, and not against an explicit j5_tile_rows
        if (rows > 0 && !fj && !pj && tuning("j5_tile_rows", 0) < 1) R = rows;
    }
    const int strips = (y1 - y0 + R) / R;
    const unsigned grid = (unsigned)(((long)nxw * strips + tpb - 1) / tpb);
    // Non-temporal stores of `out` (j5_nt_stores: 1 on, 0 off, -1 = by size).  Measured, same process, planned
    // shapes: 16384^2 0.787-0.793 of peak against 0.760-0.780, 8192^2 0.782 against 0.758-0.773, but 4096^2
    // 0.729 against 0.747-0.761 -- two 134 MB arrays ping-pong through the 256 MB Infinity Cache, and a store
    // that bypasses it takes the next step's input away.  So: on from 150 MB per array (nt_stores_for).
    const int nts = nt_stores_for(ld, y0, y1);
    if (pj) {   // the peer transport's launch: as the framed one below
        if constexpr (VEC == 2 && !NT) {
            const long cells = 2L * (pj->fx1 - pj->fx0 + 1) + 2L * (pj->fy1 - pj->fy0 + 1);
            long nb = ((cells + 64 * tpb - 1) / (64 * tpb) + 7) & ~7L;
            pj->nblocks = (int)(nb < 8 ? 8 : nb > 256 ? 256 : nb);
            pj->nunb = pj->nun > 0 ? 16 : 0;             // join workgroups: two per strip at eight strips, a multiple of 8
            if (nts)
                hipLaunchKernelGGL((jacobi5_tile_peer<2, 2, 2>), dim3(grid + pj->nblocks + pj->nunb), dim3(64 * tpb), 0, s, in, out, ld,
                                   x0, x1, y0, y1, c_first, nxw, flags, *pj);
            else
                hipLaunchKernelGGL((jacobi5_tile_peer<2, 2, 0>), dim3(grid + pj->nblocks + pj->nunb), dim3(64 * tpb), 0, s, in, out, ld,
                                   x0, x1, y0, y1, c_first, nxw, flags, *pj);
        }
        return;
    }
    if (fj) {
        // frame workgroups first (they are dispatched first): a multiple of 8 of them, so that the tile
        // workgroups keep the XCD each would have had in the plain launch (round-robin dealing)
        if constexpr (VEC == 2 && !NT) {
            const long cells = 2L * (fj->fx1 - fj->fx0 + 1) + 2L * (fj->fy1 - fj->fy0 + 1);
            long nb = ((cells + 64 * tpb - 1) / (64 * tpb) + 7) & ~7L;
            fj->nblocks = (int)(nb < 8 ? 8 : nb > 256 ? 256 : nb);
            if (nts)
                hipLaunchKernelGGL((jacobi5_tile_framed<2, 2, 2>), dim3(grid + fj->nblocks), dim3(64 * tpb), 0, s, in,
                                   out, ld, x0, x1, y0, y1, c_first, nxw, flags, *fj);
            else
                hipLaunchKernelGGL((jacobi5_tile_framed<2, 2, 0>), dim3(grid + fj->nblocks), dim3(64 * tpb), 0, s, in,
                                   out, ld, x0, x1, y0, y1, c_first, nxw, flags, *fj);
        }
        return;
    }
    // (capping the resident waves with unused LDS -- 32 down to 16 waves per CU -- changes nothing
    // until 16, where it costs 2 %: the band of rows in flight is not a lever)
    if constexpr (VEC == 2 && !NT) {
        if (nts && (R == 2 || R == 3)) {                 // the two heights the planner chooses from
            if (R == 2)
                hipLaunchKernelGGL((jacobi5_tile<2, 2, 2>), dim3(grid), dim3(64 * tpb), 0, s, in, out, ld, x0, x1, y0, y1,
                                   c_first, nxw, flags);
            else
                hipLaunchKernelGGL((jacobi5_tile<2, 3, 2>), dim3(grid), dim3(64 * tpb), 0, s, in, out, ld, x0, x1, y0, y1,
                                   c_first, nxw, flags);
            return;
        }
    }
#define DLESM_TILE(RR)                                                                               \
    hipLaunchKernelGGL((jacobi5_tile<VEC, RR, (NT ? 3 : 0)>), dim3(grid), dim3(64 * tpb), 0, s, in, out, ld, x0, x1, \
                       y0, y1, c_first, nxw, flags)
#ifdef DLESM_LAB
    switch (R) {
    case 1: DLESM_TILE(1); break;
    case 2: DLESM_TILE(2); break;
    case 3: DLESM_TILE(3); break;
    case 4: DLESM_TILE(4); break;
    case 6: DLESM_TILE(6); break;
    case 12: DLESM_TILE(12); break;
    case 16: DLESM_TILE(16); break;
    default: DLESM_TILE(8); break;
    }
#else
    if constexpr (VEC == 2) {
        if (R == 3) DLESM_TILE(3);
        else DLESM_TILE(2);
    } else {
        DLESM_TILE(4);
    }
#endif
#undef DLESM_TILE
}

// The one-cell-wide frame of the box (rows ystart/ystop, columns xstart/xstop):
// the cells whose values neighbours need first in the distributed step.
__global__ __launch_bounds__(256) void jacobi5_frame(const double *__restrict__ in,
                                                     double *__restrict__ out, int ld, int x0, int x1,
                                                     int y0, int y1, FramePack pk)
{
    const long total = frame_cells(x1 - x0 + 1, y1 - y0 + 1);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long)gridDim.x * blockDim.x)
        frame_cell<false>(t, in, out, ld, x0, x1, y0, y1, pk);
}

// The other two-row wave-tile sweeps (3 x 3, masked, continuity) have the Jacobi sweep's tile geometry, and what decides the
// launch shape is the geometry (which XCD the tile below lands on), not the arithmetic: a shape the planning call
// (dlesm_stencil5_autotune_f64) measured for this (pitch, box) serves them too -- +1.7 / +2.5 points for the 3 x 3 / masked
// sweeps at 16384^2, nothing lost elsewhere (scripts/shape_share_probe.py); without a plan, the rule.
void shape_for_tile_sweep(int ld, int x0, int x1, int y0, int y1, int *nxw_io, int *tpb_out)
{
    if (tuning("j5_autoshape", 1) && tuning("j5_use_tuned", 1) && !tuning("j5_tpb", 0)) {
        std::lock_guard<std::mutex> lk(g_shape_mu);
        auto it = g_shape_cache.find(ShapeKey{ld, x0, x1, y0, y1, 2});
        if (it != g_shape_cache.end() && it->second.nxw >= *nxw_io) {
            *nxw_io = it->second.nxw;
            *tpb_out = it->second.tpb;
            return;
        }
    }
    choose_block_shape(nxw_io, tpb_out);
}

int check_box(const char *who, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
              int ring)
{
    if (ld < 1 || ny < 1) return fail(DLESM_EINVAL, "%s: array extents %dx%d", who, ld, ny);
    if (xstart - ring < 1 || xstop + ring > ld || ystart - ring < 1 || ystop + ring > ny)
        return fail(DLESM_EINVAL, "%s: box (%d:%d,%d:%d) with a %d-cell stencil ring does not fit in "
                                  "an array of %dx%d", who, xstart, xstop, ystart, ystop, ring, ld, ny);
    return DLESM_OK;
}

#ifdef DLESM_LAB
template <int VEC, int U, bool NT, bool PIPE>
static void launch_march_u(const double *in, double *out, int ld, int x0, int x1, int y0, int y1,
                           int rows, int flags, hipStream_t s)
{
    const int c_first = x0 / VEC, c_last = x1 / VEC;
    const int nch = c_last - c_first + 1;
    const int nxb = (nch + 255) / 256;
    const int nstrips = (y1 - y0 + rows) / rows;
    dim3 grid((unsigned)(nxb * nstrips)), block(256);
    hipLaunchKernelGGL((jacobi5_march<VEC, U, NT, PIPE>), grid, block, 0, s, in, out, ld, x0, x1, y0, y1,
                       c_first, nxb, rows, flags);
}

template <int VEC, bool NT, bool PIPE>
static void launch_march(const double *in, double *out, int ld, int x0, int x1, int y0, int y1,
                         int rows, int flags, int unroll, hipStream_t s)
{
    if (unroll == 2) launch_march_u<VEC, 2, NT, PIPE>(in, out, ld, x0, x1, y0, y1, rows, flags, s);
    else if (unroll == 8) launch_march_u<VEC, 8, NT, PIPE>(in, out, ld, x0, x1, y0, y1, rows, flags, s);
    else launch_march_u<VEC, 4, NT, PIPE>(in, out, ld, x0, x1, y0, y1, rows, flags, s);
}

// Strip height when the caller does not fix it: give every resident block slot
// (8 blocks of 256 threads per CU) exactly one strip, so that the whole grid is
// co-resident, all blocks stream for the whole duration of the launch (no tail
// round) and the re-read of the two halo rows per strip is amortised over as
// many rows as possible.
static int auto_rows(int ncols_vec, int height)
{
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        slots = cus * 8;
    }
    const int nxb = (ncols_vec + 255) / 256;
    int nstrips = slots / nxb;
    if (nstrips < 1) nstrips = 1;
    int rows = (height + nstrips - 1) / nstrips;
    return rows < 1 ? 1 : rows;
}

#endif // DLESM_LAB

int launch_stencil5(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart,
                    int ystop, hipStream_t s)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK; // empty box: a zero-trip loop nest
    if (int rc = check_box("dlesm_stencil5_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && in != out, "stencil5: null or aliased arrays");
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    // tuning bits -- 1: non-temporal loads/stores, 2: no register double buffering,
    // 4: force VEC=1, 8: diagnostic "no edge loads" (wrong results, profiling only)
    // (bits 1, 2 and 8 select comparison forms / a diagnostic: libdlesm_hip_lab.so only)
    const int variant = tuning("j5_variant", 0) & (kLab ? ~0 : 0x14);
    const int flags = (variant >> 3) & 1;
#ifdef DLESM_LAB
    const bool nt = variant & 1;
    const bool pipe = !(variant & 2);
    const int unroll = tuning("j5_unroll", 4);
#endif
    // 16-byte lanes are used on an odd leading dimension too (DL_ESM_ALIGNMENT unset or odd):
    // every other row is then only 8-byte aligned, which global_load/store_dwordx4 accept --
    // 72 % of HBM peak at 16384^2 against 67 % with 8-byte lanes (variant bit 16 turns it off).
    // Only when the box's east ring column is still inside the last whole 2-column chunk.
    const bool odd_ok = !(variant & 16) && x1 + 1 <= 2 * (ld / 2) - 1;
    const bool vec2 = !(variant & 4) && ((ld % 2 == 0) || odd_ok) && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    if (tuning("j5_kernel", 0) == 0) { // XCD band sweep (default)
        // rows per tile: 2 for 16-byte lanes, 4 for the 8-byte-lane fallback (measured, scripts/size_probe.py)
        int R = tuning("j5_tile_rows", 0);
        if (R < 1) R = vec2 ? 2 : 4;
#ifdef DLESM_LAB
        if (nt) {
            if (vec2) launch_tile<2, true>(in, out, ld, x0, x1, y0, y1, R, flags, s);
            else launch_tile<1, true>(in, out, ld, x0, x1, y0, y1, R, flags, s);
            DLESM_HIP_TRY(hipGetLastError());
            return DLESM_OK;
        }
#endif
        if (vec2) launch_tile<2, false>(in, out, ld, x0, x1, y0, y1, R, flags, s);
        else launch_tile<1, false>(in, out, ld, x0, x1, y0, y1, R, flags, s);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
#ifndef DLESM_LAB
    return fail(DLESM_EINVAL, "stencil5: no such kernel in this library");      // (unreachable: j5_kernel reads 0 here)
#else
    if (tuning("j5_kernel", 0) == 3 && vec2)   // the fused-step kernel's tile shape with one step
        return launch_stencil5_multi(in, out, ld, ny, 1, xstart, xstop, ystart, ystop, xstart, xstop, ystart, ystop,
                                     0, 0, 0, 0, s);
    if (tuning("j5_kernel", 0) == 2 && vec2) { // LDS-staged comparison kernel
        int tpb = tuning("j5_tpb", 4), R = tuning("j5_tile_rows", 8);
        if (tpb != 1 && tpb != 2 && tpb != 4 && tpb != 8 && tpb != 16) tpb = 4;
        if (R != 2 && R != 4 && R != 16) R = 8;
        while ((size_t)(R + 2) * (128 * tpb + 4) * sizeof(double) > 65536) tpb /= 2;   // 64 KiB of LDS per group
        const int nt = 64 * tpb, c_first = (x0 / 2) & ~7, c_last = x1 / 2;
        int nxb = (c_last - c_first + nt) / nt;
        if (tuning("j5_autoshape", 1) && nxb > 8) nxb = (nxb + 7) & ~7;   // tile below: same XCD
        nxb += tuning("j5_pad_tiles", 0);
        const unsigned grid = (unsigned)((long)nxb * ((y1 - y0 + R) / R));
        const size_t lds = (size_t)(R + 2) * (2 * nt + 4) * sizeof(double);
#define DLESM_LDS(RR) hipLaunchKernelGGL(jacobi5_lds<RR>, dim3(grid), dim3(nt), lds, s, in, out, ld, x0, x1, y0, y1, c_first, nxb)
        switch (R) {
        case 2: DLESM_LDS(2); break;
        case 4: DLESM_LDS(4); break;
        case 16: DLESM_LDS(16); break;
        default: DLESM_LDS(8); break;
        }
#undef DLESM_LDS
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    int rows = tuning("j5_rows", 0);
    if (rows < 1) rows = auto_rows(vec2 ? x1 / 2 - x0 / 2 + 1 : x1 - x0 + 1, y1 - y0 + 1);
#define DLESM_J5(V, N, P) launch_march<V, N, P>(in, out, ld, x0, x1, y0, y1, rows, flags, unroll, s)
    if (vec2) {
        if (nt) { if (pipe) DLESM_J5(2, true, true); else DLESM_J5(2, true, false); }
        else { if (pipe) DLESM_J5(2, false, true); else DLESM_J5(2, false, false); }
    } else {
        if (nt) { if (pipe) DLESM_J5(1, true, true); else DLESM_J5(1, true, false); }
        else { if (pipe) DLESM_J5(1, false, true); else DLESM_J5(1, false, false); }
    }
#undef DLESM_J5
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
#endif // DLESM_LAB
}

// One wave parked on the frame flag: lane 0 sleeps until the flag reaches `seq`.  It holds one wave
// slot and issues one load per sleep period; it never spins hot.  The wait is
// bounded (about 20 s of the 100 MHz real-time counter): if the frame never reports -- which only a
// failed launch could cause -- the kernel gives up, raises *timed_out (pinned host memory, checked
// by the next step call) and lets the stream drain instead of hanging the device.
__global__ void frame_flag_wait(const unsigned long long *flag, unsigned long long seq, int *timed_out,
                                unsigned long long max_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // relaxed polls, ONE acquire behind them (polling with acquire loads is the guide's "correct, 2-3x slower per hop");
    // whatever this stream runs next starts behind a kernel boundary anyway
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(64);
        if (max_ticks && __builtin_amdgcn_s_memrealtime() - t0 > max_ticks) {        // 100 MHz counter; 0 = no limit
            __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    handover_acquire<false>();
}

__global__ void flag_set(unsigned long long *flag, unsigned long long seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_flag_set(unsigned long long *flag, unsigned long long seq, hipStream_t s)
{
    hipLaunchKernelGGL(flag_set, dim3(1), dim3(64), 0, s, flag, seq);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

unsigned long long remote_wait_ticks()
{
    const int sec = tuning("dm_wait_seconds", 600);
    return sec <= 0 ? 0ull : (unsigned long long)sec * 100000000ull;
}

int launch_frame_flag_wait(const unsigned long long *flag, unsigned long long seq, int *timed_out, hipStream_t s, bool remote)
{
    // local: this GPU's own frame workgroups report within microseconds -- 30 s means a failed launch.  remote: the flag
    // follows an exchange, i.e. it waits for the slowest neighbour (a rank that is writing output, say), which the
    // reference does without limit (MPI_Waitany, parallel_comms_mod.f90:1773-1798): dm_wait_seconds, default 10 minutes.
    hipLaunchKernelGGL(frame_flag_wait, dim3(1), dim3(64), 0, s, flag, seq, timed_out, remote ? remote_wait_ticks() : 3000000000ull);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// Do kernels of two streams really run side by side in this process?  The one-launch and time-loop forms
// of the distributed step park a waiting kernel on the side stream while the kernel that will release it
// runs on the caller's stream; under a tool that serialises kernel execution (rocprofv3 --pmc,
// AMD_SERIALIZE_KERNEL) the waiter would run alone and only leave through its time-out (seen: a --pmc
// run of the loop-back script crawled from time-out to time-out).  Probe once per process: a waiter
// with a 50 ms bound on the side stream, the releasing store on the caller's stream.
static std::mutex g_probe_mu;
static std::map<hipStream_t, int> g_probe_seen;

void invalidate_concurrency_probe()
{
    std::lock_guard<std::mutex> lk(g_probe_mu);
    g_probe_seen.clear();
}

bool streams_run_concurrently(hipStream_t callers)
{
    // per caller's stream: which hardware queue a stream lands on is the runtime's choice.  The answer is kept until it
    // is invalidated: dlesm_probe_stream_concurrency (the host program re-created a stream, attached a tool), an
    // acknowledged wait time-out, dlesm_finalize.  The FIRST call for a stream synchronises the device once (so that
    // the probe's two kernels start together), allocates 64 bytes and waits up to 50 ms: do it outside graph captures
    // (the steps ask for it only when they are not being captured) -- dlesm_halo_plan_create does it for the null stream.
    std::mutex &mu = g_probe_mu;
    std::map<hipStream_t, int> &seen = g_probe_seen;
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(callers);
    if (it != seen.end()) return it->second != 0;
    int &cached = seen[callers];
    cached = 0;
    unsigned long long *flag = nullptr;
    int *timed_out = nullptr;
    if (hipMalloc((void **)&flag, 64) != hipSuccess) return false;
    if (hipHostMalloc((void **)&timed_out, sizeof(int), hipHostMallocMapped) != hipSuccess) {
        (void)hipFree(flag);
        return false;
    }
    *timed_out = 0;
    hipStream_t a = side_stream(), b = callers;          // the two streams the step itself will use
    // one-time: drain everything first, so that the probe's two kernels start at once
    bool ok = hipMemset(flag, 0, 64) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(frame_flag_wait, dim3(1), dim3(64), 0, a, flag, 1ull, timed_out, 5000000ull);   // 50 ms
        hipLaunchKernelGGL(flag_set, dim3(1), dim3(64), 0, b, flag, 1ull);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(a) == hipSuccess &&
             hipStreamSynchronize(b) == hipSuccess;
    }
    if (ok && *(volatile int *)timed_out == 0) cached = 1;
    (void)hipFree(flag);
    (void)hipHostFree(timed_out);
    return cached != 0;
}

int launch_stencil5_framed(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart,
                           int ystop, FrameJob job, hipStream_t s, bool *fused)
{
    *fused = false;
    if (xstop - xstart < 2 || ystop - ystart < 2) return DLESM_OK;          // no interior: two-launch path
    if (int rc = check_box("dlesm_jacobi5_step_dm", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && in != out, "stencil5: null or aliased arrays");
    DLESM_REQUIRE(job.counter != nullptr && job.flag != nullptr, "stencil5 framed: no signal words");
    const int variant = tuning("j5_variant", 0) & (kLab ? ~0 : 0x14);
    const int x1i = xstop - 2;                                              // east end of the interior, 0-based
    const bool odd_ok = !(variant & 16) && x1i + 1 <= 2 * (ld / 2) - 1;
    const bool vec2 = !(variant & 4) && ((ld % 2 == 0) || odd_ok) && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    int R = tuning("j5_tile_rows", 0);
    if (R < 1) R = 2;
    if (!vec2 || (variant & 1) || tuning("j5_kernel", 0) != 0 || R != 2) return DLESM_OK;
    job.fx0 = xstart - 1, job.fx1 = xstop - 1, job.fy0 = ystart - 1, job.fy1 = ystop - 1;
    launch_tile<2, false>(in, out, ld, xstart, xstop - 2, ystart, ystop - 2, 2, (variant >> 3) & 1, s, &job);
    DLESM_HIP_TRY(hipGetLastError());
    *fused = true;
    return DLESM_OK;
}

int launch_stencil5_peer(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         PeerJob job, hipStream_t s, bool *fused)
{
    *fused = false;
    if (xstop - xstart < 2 || ystop - ystart < 2) return DLESM_OK;          // no interior
    if (int rc = check_box("dlesm_jacobi5_step_dm", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(in != nullptr && out != nullptr && in != out, "stencil5: null or aliased arrays");
    DLESM_REQUIRE(job.counter != nullptr && job.nin <= PeerJob::MAXM && job.nout <= PeerJob::MAXM, "stencil5 peer: bad job");
    const int variant = tuning("j5_variant", 0) & (kLab ? ~0 : 0x14);
    const int x1i = xstop - 2;
    const bool odd_ok = !(variant & 16) && x1i + 1 <= 2 * (ld / 2) - 1;
    const bool vec2 = !(variant & 4) && ((ld % 2 == 0) || odd_ok) && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    int R = tuning("j5_tile_rows", 0);
    if (R < 1) R = 2;
    if (!vec2 || (variant & 1) || tuning("j5_kernel", 0) != 0 || R != 2) return DLESM_OK;
    job.fx0 = xstart - 1, job.fx1 = xstop - 1, job.fy0 = ystart - 1, job.fy1 = ystop - 1;
    launch_tile<2, false>(in, out, ld, xstart, xstop - 2, ystart, ystop - 2, 2, (variant >> 3) & 1, s, nullptr, &job);
    DLESM_HIP_TRY(hipGetLastError());
    *fused = true;
    return DLESM_OK;
}

int launch_stencil5_peer_frame(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                               PeerJob job, hipStream_t s)
{
    if (xstop < xstart || ystop < ystart) return launch_peer_flags_set(nullptr, 0, job.seq, job.seqw, job.timed_out, s);
    if (int rc = check_box("dlesm_jacobi5_step_dm", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    job.fx0 = xstart - 1, job.fx1 = xstop - 1, job.fy0 = ystart - 1, job.fy1 = ystop - 1;
    const long cells = 2L * (xstop - xstart + 1) + 2L * (ystop - ystart + 1);
    long nb = (cells + 255) / 256;
    job.nblocks = (int)(nb < 1 ? 1 : nb > 256 ? 256 : nb);
    job.nun = job.nunb = 0;                               // (the join of this form is the separate launch)
    // the frame workgroups of the one-launch kernel with no tile workgroup behind them
    hipLaunchKernelGGL((jacobi5_tile_peer<2, 2, 0>), dim3(job.nblocks), dim3(256), 0, s, in, out, ld, 0, 0, 0, 0, 0, 1, 0, job);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

int launch_peer_unpack(const PeerStrips &st, unsigned long long seq, const unsigned long long *seqw, double *const *fields, int nf, int ld,
                       int *timed_out, hipStream_t s)
{
    if (st.n == 0) return DLESM_OK;
    DLESM_REQUIRE(nf >= 1 && nf <= 16, "peer unpack of %d fields", nf);
    long longest = 1;
    for (int k = 0; k < st.n; k++) longest = std::max(longest, (long)st.s[k].ni * st.s[k].nj);
    int parts = (int)((longest + 255) / 256);
    if (parts > 16) parts = 16;
    PeerFields pf{};
    for (int k = 0; k < nf; k++) pf.f[k] = fields[k];
    hipLaunchKernelGGL(peer_unpack_k, dim3(parts, st.n, nf), dim3(256), 0, s, st, seq, seqw, pf, ld, remote_wait_ticks(), timed_out,
                       tuning("mailbox_fences", 1));
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

int launch_peer_pack(const PeerOuts &out, const double *const *fields, int nf, int ld, unsigned *counter, unsigned long long seq,
                     unsigned long long *seqw, int *sticky, hipStream_t s)
{
    if (out.n == 0) return launch_peer_flags_set(nullptr, 0, seq, seqw, sticky, s);    // nothing to send: the operation still counts
    DLESM_REQUIRE(nf >= 1 && nf <= 16 && counter != nullptr, "peer pack of %d fields", nf);
    long longest = 1;
    for (int k = 0; k < out.n; k++) longest = std::max(longest, (long)out.s[k].ni * out.s[k].nj);
    int parts = (int)((longest + 255) / 256);
    if (parts > 16) parts = 16;
    PeerFields pf{};
    for (int k = 0; k < nf; k++) pf.f[k] = const_cast<double *>(fields[k]);
    hipLaunchKernelGGL(peer_pack_k, dim3(parts, out.n, nf), dim3(256), 0, s, out, pf, ld, counter, seq, seqw, sticky, tuning("mailbox_fences", 1));
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

int launch_peer_exchange(const PeerOuts &out, const PeerStrips &in, double *const *fields, int nf, int ld, unsigned *counter,
                         unsigned long long seq, unsigned long long *seqw, int *timed_out, hipStream_t s)
{
    if (out.n == 0 && in.n == 0) return launch_peer_flags_set(nullptr, 0, seq, seqw, timed_out, s);   // the operation still counts
    DLESM_REQUIRE(nf >= 1 && nf <= 16 && counter != nullptr, "peer exchange of %d fields", nf);
    long longest = 1;
    for (int k = 0; k < out.n; k++) longest = std::max(longest, (long)out.s[k].ni * out.s[k].nj);
    for (int k = 0; k < in.n; k++) longest = std::max(longest, (long)in.s[k].ni * in.s[k].nj);
    int parts = (int)((longest + 255) / 256);
    if (parts > 16) parts = 16;
    PeerFields pf{};
    for (int k = 0; k < nf; k++) pf.f[k] = fields[k];
    hipLaunchKernelGGL(peer_exchange_k, dim3(parts, std::max(out.n, in.n), 2 * nf), dim3(256), 0, s, out, in, pf, nf, ld, counter,
                       seq, seqw, remote_wait_ticks(), timed_out, tuning("mailbox_fences", 1));
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

int launch_peer_flags_set(unsigned long long *const *flags, int n, unsigned long long seq, unsigned long long *seqw, int *sticky,
                          hipStream_t s)
{
    if (n == 0 && !seqw) return DLESM_OK;                 // (n == 0 with sequence words: the launch only moves them on)
    DLESM_REQUIRE(n <= PeerJob::MAXM, "%d peer flags", n);
    PeerFlagList fl{};
    for (int k = 0; k < n; k++) fl.f[k] = flags[k];
    hipLaunchKernelGGL(peer_flags_set_k, dim3(1), dim3(64), 0, s, fl, n, seq, seqw, sticky, tuning("mailbox_fences", 1));
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

int launch_stencil5_frame(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                          int ystart, int ystop, hipStream_t s, const FramePack *pack)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("stencil5 frame", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    const long cells = 2L * (xstop - xstart + 1) + 2L * (ystop - ystart + 1);
    int blocks = (int)((cells + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    FramePack pk{};
    if (pack) pk = *pack;
    hipLaunchKernelGGL(jacobi5_frame, dim3(blocks), dim3(256), 0, s, in, out, ld, xstart - 1, xstop - 1,
                       ystart - 1, ystop - 1, pk);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// ===========================================================================
// Shallow-water u/v/h update, NE staggering, all intermediates (cu, cv, z, h)
// recomputed in registers from the 3x3 neighbourhood of u, v, p: 6 fields read,
// 3 written = 72 B/cell of algorithmic traffic.
// First, direct form: neighbours come from L1/L2.
// ===========================================================================
__global__ __launch_bounds__(256) void shallow_step_direct(
    dlesm_sw_params q, int ld, int x0, int x1, int y0, int y1, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ p, const double *__restrict__ uold,
    const double *__restrict__ vold, const double *__restrict__ pold, double *__restrict__ unew,
    double *__restrict__ vnew, double *__restrict__ pnew)
{
    const int i = x0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > x1) return;
    for (int j = y0 + blockIdx.y; j <= y1; j += gridDim.y)
        shallow_point_ne(q, ld, (size_t)j * ld + i, u, v, p, uold, vold, pold, unew, vnew, pnew);
}

// the same for boxes a few columns wide and many rows tall (frame columns): lanes run along j
__global__ __launch_bounds__(256) void shallow_step_direct_cols(
    dlesm_sw_params q, int ld, int x0, int x1, int y0, int y1, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ p, const double *__restrict__ uold,
    const double *__restrict__ vold, const double *__restrict__ pold, double *__restrict__ unew,
    double *__restrict__ vnew, double *__restrict__ pnew)
{
    const int j = y0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j > y1) return;
    for (int i = x0; i <= x1; i++)
        shallow_point_ne(q, ld, (size_t)j * ld + i, u, v, p, uold, vold, pold, unew, vnew, pnew);
}

// the one-cell frame of the box (the cells a neighbour needs first in the distributed step): one cell
// per thread, all four sides in ONE launch; cells a neighbour will receive also go into the aggregated
// send buffer of the three new fields (FramePack3)
__global__ __launch_bounds__(256) void shallow_frame_k(
    dlesm_sw_params q, int ld, int x0, int x1, int y0, int y1, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ p, const double *uold,
    const double *vold, const double *pold, double *__restrict__ unew,
    double *__restrict__ vnew, double *__restrict__ pnew, FramePack3 pk, int smooth, double alpha)
{
    const long total = frame_cells(x1 - x0 + 1, y1 - y0 + 1);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int i, j;
        frame_index(t, x0, x1, y0, y1, i, j);
        const size_t o = (size_t)j * ld + i;
        const SwPoint r = shallow_values_ne(q, ld, o, u, v, p, uold, vold, pold);
        unew[o] = r.un;
        vnew[o] = r.vn;
        pnew[o] = r.pn;
        if (smooth) smooth_old_level(alpha, o, u, v, p, r, const_cast<double *>(uold), const_cast<double *>(vold), const_cast<double *>(pold));
        for (int k = 0; k < pk.n; k++)
            if (pk.holds(k, i, j)) {
                double *b = pk.at(k);          // (a neighbour's mailbox with the peer transport: uncached memory, the
                b[pk.slot(k, 0, i, j)] = r.un; //  stores have left by the end of the kernel; its flags are raised by a
                b[pk.slot(k, 1, i, j)] = r.vn; //  launch behind this one)
                b[pk.slot(k, 2, i, j)] = r.pn;
            }
    }
}

int launch_shallow_frame(const dlesm_sw_params &q, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         const double *u, const double *v, const double *p, const double *uold,
                         const double *vold, const double *pold, double *unew, double *vnew, double *pnew,
                         const FramePack3 *pack, hipStream_t s, const double *smooth_alpha)
{
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("shallow frame", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    const long cells = 2L * (xstop - xstart + 1) + 2L * (ystop - ystart + 1);
    int blocks = (int)((cells + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    FramePack3 pk{};
    if (pack) pk = *pack;
    hipLaunchKernelGGL(shallow_frame_k, dim3(blocks), dim3(256), 0, s, q, ld, xstart - 1, xstop - 1, ystart - 1,
                       ystop - 1, u, v, p, uold, vold, pold, unew, vnew, pnew, pk, smooth_alpha ? 1 : 0,
                       smooth_alpha ? *smooth_alpha : 0.0);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// ===========================================================================
// small utility kernels
// ===========================================================================
__global__ void copy_patch_k(const double *__restrict__ src, double *__restrict__ dst, int ld, int sx0,
                             int sy0, int dx0, int dy0, int nx, int ny)
{
    for (int j = blockIdx.y; j < ny; j += gridDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += gridDim.x * blockDim.x)
            dst[(size_t)(dy0 + j) * ld + dx0 + i] = src[(size_t)(sy0 + j) * ld + sx0 + i];
}

// whole rows of a field are one contiguous block: the linear copy that sets the measured ceiling
// (scripts/membench.hip), one 16-byte element per thread, workgroups sweeping memory front to back
template <bool NTS>
__global__ __launch_bounds__(256) void copy_linear_k(const double *__restrict__ src, double *__restrict__ dst, size_t n2)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    if (NTS) __builtin_nontemporal_store(((const d2 *)src)[i], (d2 *)dst + i);
    else ((d2 *)dst)[i] = ((const d2 *)src)[i];
}

__global__ void fill_k(double *__restrict__ f, int ld, int x0, int y0, int nx, int ny, double value)
{
    for (int j = blockIdx.y; j < ny; j += gridDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += gridDim.x * blockDim.x)
            f[(size_t)(y0 + j) * ld + x0 + i] = value;
}

// Round 3: the utility sweeps as ROW SEGMENTS (dlesm_device.h); the forms above remain for unaligned base pointers and
// rows of a few elements.  The earlier forms gave every workgroup whole rows (thousands of concurrent row fronts, 8-byte
// lanes): 64-71 % of the HBM peak at 16384^2 (profiles/r02_aux_kernels.txt).
typedef rs_d2 d2u;

// the same value into n2 16-byte elements of contiguous memory: whole rows (set_field of a field) are one block
__global__ __launch_bounds__(256) void fill_linear_k(double *__restrict__ f, size_t n2, double value, bool nt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    if (nt) __builtin_nontemporal_store(d2u{value, value}, (d2u *)f + i);
    else ((d2u *)f)[i] = d2u{value, value};
}

template <class GEN>   // GEN(i, j) -> value of element (0-based column i, row j)
__global__ __launch_bounds__(256) void rowseg_write_k(double *__restrict__ f, int ld, int x0, int y0, int nx, int segs, int segp,
                                                      bool nt, GEN gen)
{
    const int jr = blockIdx.x / segs, sg = blockIdx.x - jr * segs, j = y0 + jr;
    const long row = (long)j * ld, e0 = row + x0, e1 = e0 + nx - 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const RowPair pr = rowseg_pair(e0, e1, sg, segp, threadIdx.x, k);
        if (!pr.any()) continue;
        const int i = (int)(pr.el - row);
        if (pr.full()) {
            const d2u v = d2u{gen(i, j), gen(i + 1, j)};
            if (nt) __builtin_nontemporal_store(v, (d2u *)(f + pr.el));
            else *(d2u *)(f + pr.el) = v;
        } else if (pr.m0) f[pr.el] = gen(i, j);
        else f[pr.el + 1] = gen(i + 1, j);
    }
}
// Round 3, measured with a stand-alone program (scripts/store_probe.hip, 16448 x 16387 array, write-only, non-temporal): a
// LINEAR sweep over whole rows that merely masks the columns outside the box runs at 83.1 % of the HBM peak (the plain linear
// fill: 84.2 %), the (row, segment) decomposition above at 75.7 % with 256 pairs per workgroup and 68.8 % with 1024 (four
// stores per thread 4 KiB apart) -- workgroups whose 4 KiB chunk is aligned in MEMORY, all of them full, is what the store
// stream wants; a box that covers half of each row reaches 64 % however it is indexed (the gaps cost).  So: boxes that cover
// most of the row pitch (>= 3/4) of an even-pitch array take the linear form, thread t <-> pair t of rows y0 .. y0+h-1.
template <class GEN>
__global__ __launch_bounds__(256) void rowlinear_write_k(double *__restrict__ f, int ld, int x0, int y0, int nx, size_t n2, bool nt,
                                                         GEN gen)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n2) return;
    const int pr = ld >> 1;
    const int jr = (int)(t / pr), c = (int)(t - (size_t)jr * pr), j = y0 + jr, i = 2 * c;
    const bool m0 = i >= x0 && i < x0 + nx, m1 = i + 1 >= x0 && i + 1 < x0 + nx;
    if (!m0 && !m1) return;
    double *p = f + (size_t)j * ld + i;
    if (m0 && m1) {
        const d2u v = d2u{gen(i, j), gen(i + 1, j)};
        if (nt) __builtin_nontemporal_store(v, (d2u *)p);
        else *(d2u *)p = v;
    } else if (m0) p[0] = gen(i, j);
    else p[1] = gen(i + 1, j);
}
// the linear form applies: even pitch, 16-byte aligned base, the box covers at least 3/4 of the pitch
static bool rowlinear_ok(const double *f, int ld, int nx, int nyb)
{
    return ld % 2 == 0 && (uintptr_t)f % 16 == 0 && 4L * nx >= 3L * ld && ((size_t)(ld / 2) * nyb + 255) / 256 < ((size_t)1 << 31) &&
           tuning("util_rowlinear", 1);
}

struct GenConst {
    double v;
    __device__ double operator()(int, int) const { return v; }
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__global__ void hash_init_k(double *__restrict__ f, int ld, int x0, int y0, int nx, int ny,
                            uint64_t seed, int64_t gx0, int64_t gy0)
{
    for (int j = blockIdx.y; j < ny; j += gridDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += gridDim.x * blockDim.x) {
            const uint64_t gi = (uint64_t)(gx0 + x0 + i), gj = (uint64_t)(gy0 + y0 + j);
            const uint64_t h = splitmix64(seed ^ (gi + (gj << 32)));
            f[(size_t)(y0 + j) * ld + x0 + i] = (double)(h >> 11) * 0x1.0p-53;
        }
}

struct GenHash {
    uint64_t seed;
    int64_t gx0, gy0;
    __device__ double operator()(int i, int j) const
    {
        const uint64_t gi = (uint64_t)(gx0 + i), gj = (uint64_t)(gy0 + j);
        return (double)(splitmix64(seed ^ (gi + (gj << 32))) >> 11) * 0x1.0p-53;
    }
};

// SUM(ABS()) in a fixed tree order: lane strides -> wave shuffle tree -> LDS
// across the 4 waves -> one partial per block -> second pass over the partials.
__device__ __forceinline__ double block_sum_256(double v)
{
    __shared__ double wsum[4];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    __syncthreads();
    return r;
}

// SUM(ABS(f(box))) partials, one per workgroup: rows dealt round-robin to the workgroups, 16-byte
// lanes on the even-aligned part of each row (the element before and after it by lane 0), two
// independent accumulators per lane.  The order of the additions is fixed by the launch shape, so
// the result is deterministic run to run.
__global__ __launch_bounds__(256) void abs_sum_rows(const double *__restrict__ f, int ld, int x0, int y0,
                                                    int nx, int ny, double *__restrict__ partial)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    double acc0 = 0.0, acc1 = 0.0;
    const bool base_ok = ((uintptr_t)f & 15) == 0;
    for (int j = blockIdx.x; j < ny; j += gridDim.x) {
        const size_t off = (size_t)(y0 + j) * ld + x0;
        const double *r = f + off;
        if (base_ok) {
            const int head = (int)(off & 1);                  // one element before the 16-byte aligned part
            const int nvec = (nx - head) / 2, tail = (nx - head) & 1;
            const d2 *rv = (const d2 *)(r + head);
            if (threadIdx.x == 0) {
                if (head && nx > 0) acc0 += fabs(r[0]);
                if (tail) acc1 += fabs(r[nx - 1]);
            }
            int i = threadIdx.x;
            for (; i + 768 < nvec; i += 1024) {               // four 16-byte loads in flight per lane, read once
                const d2 a = __builtin_nontemporal_load(rv + i), b = __builtin_nontemporal_load(rv + i + 256);
                const d2 c = __builtin_nontemporal_load(rv + i + 512), d = __builtin_nontemporal_load(rv + i + 768);
                acc0 += fabs(a.x) + fabs(a.y);
                acc1 += fabs(b.x) + fabs(b.y);
                acc0 += fabs(c.x) + fabs(c.y);
                acc1 += fabs(d.x) + fabs(d.y);
            }
            for (; i < nvec; i += 256) {
                const d2 a = __builtin_nontemporal_load(rv + i);
                acc0 += fabs(a.x) + fabs(a.y);
            }
        } else {
            for (int i = threadIdx.x; i < nx; i += 256) acc0 += fabs(r[i]);
        }
    }
    const double acc = block_sum_256(acc0 + acc1);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// the same sum as row segments (one partial per workgroup = per segment of a row; dlesm_device.h): the additions happen
// in an order fixed by (ld, box) alone, so the result is deterministic run to run.  All four 16-byte loads of a lane are
// issued before the first use (a pair that is not wholly inside the box reads the row's first whole pair instead and is
// fixed up from scalar loads: only the two ends of a row have such pairs).
__global__ __launch_bounds__(256) void abs_sum_rowseg(const double *__restrict__ f, int ld, int x0, int y0, int nx, int segs,
                                                      int segp, double *__restrict__ partial)
{
    const int jr = blockIdx.x / segs, sg = blockIdx.x - jr * segs;
    const long row = (long)(y0 + jr) * ld, e0 = row + x0, e1 = e0 + nx - 1;
    const long safe = (e0 + 1) & ~1L;                    // a pair wholly inside the row's part of the box (nx >= 3)
    RowPair pr[4];
    d2u v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pr[k] = rowseg_pair(e0, e1, sg, segp, threadIdx.x, k);
        v[k] = __builtin_nontemporal_load((const d2u *)(f + (pr[k].full() ? pr[k].el : safe)));
    }
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        a[k] = pr[k].full() ? fabs(v[k].x) + fabs(v[k].y) : 0.0;
        if (pr[k].any() && !pr[k].full()) a[k] = fabs(f[pr[k].m0 ? pr[k].el : pr[k].el + 1]);
    }
    const double acc = block_sum_256((a[0] + a[2]) + (a[1] + a[3]));
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// one level of the tree over the partials: workgroup b sums partial[b*chunk .. (b+1)*chunk)
__global__ __launch_bounds__(256) void sum_partials_level(const double *__restrict__ partial, long n, int chunk,
                                                          double *__restrict__ out)
{
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    double acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) acc += partial[i];
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void sum_partials(const double *__restrict__ partial, int n,
                                                    double *__restrict__ result)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) *result = acc;
}

static std::mutex g_scratch_mu;
static double *g_partials = nullptr; // 4096 partials + 1 result
static const int kMaxPartials = 4096;

} // namespace dlesm

using namespace dlesm;

extern "C" int dlesm_stencil5_f64(const double *in, double *out, int ld, int ny, int xstart,
                                  int xstop, int ystart, int ystop, void *stream)
{
    if (int rc = ensure_device()) return rc;
    return launch_stencil5(in, out, ld, ny, xstart, xstop, ystart, ystop, (hipStream_t)stream);
}

// Plan-style tuning of dlesm_stencil5_f64 for one (leading dimension, box): times ~a dozen launch
// shapes around the rule's choice on the caller's own arrays (every launch is the same valid
// step in -> out), remembers the fastest for later calls with that geometry, and returns after a
// stream synchronisation.  Optional: without it the rule of choose_block_shape() is used.
extern "C" int dlesm_stencil5_autotune_f64(const double *in, double *out, int ld, int ny, int xstart,
                                           int xstop, int ystart, int ystop, void *stream)
{
    clear_error();
    if (int rc = ensure_device()) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = launch_stencil5(in, out, ld, ny, xstart, xstop, ystart, ystop, s)) return rc;   // validates, warms
    const int x0 = xstart - 1, x1 = xstop - 1, y0 = ystart - 1, y1 = ystop - 1;
    const int variant = tuning("j5_variant", 0) & (kLab ? ~0 : 0x14);
    const bool odd_ok = !(variant & 16) && x1 + 1 <= 2 * (ld / 2) - 1;
    const bool vec2 = !(variant & 4) && ((ld % 2 == 0) || odd_ok) && ((uintptr_t)in % 16 == 0) &&
                      ((uintptr_t)out % 16 == 0);
    if (tuning("j5_kernel", 0) != 0) return DLESM_OK;            // only the default kernel has shapes
    const int VEC = vec2 ? 2 : 1;
    const int c_first = (x0 / VEC) & ~(128 / (8 * VEC) - 1), c_last = x1 / VEC;
    const int nxw = (c_last - c_first + 64) / 64;
    if (nxw < 8) return DLESM_OK;                                 // thin boxes: nothing to choose
    std::vector<Shape> cand;
    auto add = [&](int tpb, int t, int rows) {
        for (const Shape &c : cand)
            if (c.tpb == tpb && c.nxw == t && c.rows == rows) return;
        cand.push_back(Shape{tpb, t, rows});
    };
    {
        int t = nxw, tpb = 4;
        choose_block_shape(&t, &tpb);
        add(tpb, t, 0);                                           // the rule's own choice first
    }
    // tile heights: 2 rows (the default) and 3 -- at 4096^2 with a 33280-byte row pitch 3 rows measured 3-4 %
    // faster, at 16384^2 2 rows; only the 16-byte-lane form, and only when the caller has not fixed it
    const bool try_rows = vec2 && tuning("j5_tile_rows", 0) < 1;
    for (int rows : {0, 3}) {
        if (rows && !try_rows) continue;
        for (int tpb : {8, 4}) {
            const int period = 8 * tpb, dmax = tpb == 8 ? 3 : 1;
            for (int k = 0; k < 2; k++) {                         // this multiple of 8 groups and the next
                const int base = (nxw / period + k) * period;
                for (int d = 1; d <= dmax; d++) {
                    if (base - d >= nxw) add(tpb, base - d, rows);
                    if (base + d >= nxw) add(tpb, base + d, rows);
                }
            }
        }
        if (rows) add(cand[0].tpb, cand[0].nxw, rows);            // the rule's shape at the other height
    }
    hipEvent_t e0, e1;
    DLESM_HIP_TRY(hipEventCreate(&e0));
    DLESM_HIP_TRY(hipEventCreate(&e1));
    // three interleaved passes over the candidates (so that clock drift hits all of them alike); a
    // trial is 4 back-to-back launches between two events, the first trial of a pass is a warm-up
    std::vector<float> best_of(cand.size(), 1e30f);
    int rc = DLESM_OK;
    for (int pass = 0; pass < 3 && !rc; pass++)
        for (size_t k = 0; k <= cand.size() && !rc; k++) {
            const Shape c = cand[k ? k - 1 : 0];
            { std::lock_guard<std::mutex> lk(g_shape_mu); g_shape_override = c; }
            (void)hipEventRecord(e0, s);
            for (int rep = 0; rep < 4 && !rc; rep++)
                rc = launch_stencil5(in, out, ld, ny, xstart, xstop, ystart, ystop, s);
            (void)hipEventRecord(e1, s);
            { std::lock_guard<std::mutex> lk(g_shape_mu); g_shape_override = Shape{0, 0, 0}; }
            if (hipEventSynchronize(e1) != hipSuccess) rc = fail(DLESM_EHIP, "autotune: event synchronisation failed");
            float ms = 0.f;
            if (!rc && k > 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best_of[k - 1]) best_of[k - 1] = ms;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    Shape best = cand[0], best2 = cand[0];               // over all tile heights / among the default height
    float best_ms = best_of[0] * 0.995f, best2_ms = best_ms;   // the rule's choice stays unless beaten by 0.5 %
    for (size_t k = 1; k < cand.size(); k++) {
        if (best_of[k] < best_ms) { best_ms = best_of[k]; best = cand[k]; }
        if (cand[k].rows == 0 && best_of[k] < best2_ms) { best2_ms = best_of[k]; best2 = cand[k]; }
    }
    std::lock_guard<std::mutex> lk(g_shape_mu);
    g_shape_cache[ShapeKey{ld, x0, x1, y0, y1, VEC}] = best;
    g_shape_cache[ShapeKey{ld, x0, x1, y0, y1, VEC + 100}] = best2;
    return DLESM_OK;
}

extern "C" int dlesm_stencil5_planned_shape(int ld, int xstart, int xstop, int ystart, int ystop, int *waves_per_group,
                                            int *tiles_per_row, int *rows_per_tile, int *nt_stores)
{
    DLESM_REQUIRE(waves_per_group && tiles_per_row && rows_per_tile && nt_stores, "null pointer");
    *waves_per_group = *tiles_per_row = *rows_per_tile = 0;
    *nt_stores = nt_stores_for(ld, ystart - 1, ystop - 1);
    std::lock_guard<std::mutex> lk(g_shape_mu);
    for (int vec : {2, 1}) {
        auto it = g_shape_cache.find(ShapeKey{ld, xstart - 1, xstop - 1, ystart - 1, ystop - 1, vec});
        if (it == g_shape_cache.end()) continue;
        *waves_per_group = it->second.tpb;
        *tiles_per_row = it->second.nxw;
        *rows_per_tile = it->second.rows ? it->second.rows : (vec == 2 ? 2 : 4);
        break;
    }
    return DLESM_OK;
}

extern "C" int dlesm_shallow_step_f64(const dlesm_sw_params *q, int ld, int ny, int xstart, int xstop,
                                      int ystart, int ystop, const double *u, const double *v,
                                      const double *p, const double *uold, const double *vold,
                                      const double *pold, double *unew, double *vnew, double *pnew,
                                      void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(q && u && v && p && uold && vold && pold && unew && vnew && pnew, "null pointer");
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("dlesm_shallow_step_f64", ld, ny, xstart, xstop, ystart, ystop, 1)) return rc;
    DLESM_REQUIRE(unew != u && unew != v && unew != p && vnew != u && vnew != v && vnew != p &&
                      pnew != u && pnew != v && pnew != p,
                  "shallow step: outputs alias the 3x3-read inputs");
    const int nx = xstop - xstart + 1, nyb = ystop - ystart + 1;
    // 16-byte lanes; on an odd leading dimension (rows alternately 8-byte aligned) only while
    // the east ring column stays inside the last whole 2-column chunk of a row
    bool aligned = ld % 2 == 0 || (xstop - 1) + 1 <= 2 * (ld / 2) - 1;
    for (const double *f : {u, v, p, uold, vold, pold, (const double *)unew, (const double *)vnew,
                            (const double *)pnew})
        aligned = aligned && ((uintptr_t)f % 16 == 0);
    // sw_kernel: 0 (default) = register-tiled sweep of dlesm_shallow.hip, 73.7 % of HBM peak at
    // 8192^2; 1 = direct form, 62 % (scripts/shallow_probe.py, profiles/r01_shallow_*.txt)
    // boxes a few columns wide (the west/east frame columns of the distributed step): a wave tile would load
    // 128 columns x 4 rows of three arrays for two useful cells -- one cell per thread moves 6 x less
    const bool thin = nx <= SW_THIN_BOX && nyb > 8;
    if (aligned && tuning("sw_kernel", 0) == 0 && !thin) {
        launch_shallow_tile(*q, ld, xstart - 1, xstop - 1, ystart - 1, ystop - 1, u, v, p, uold, vold, pold,
                            unew, vnew, pnew, (hipStream_t)stream);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    // odd leading dimension / thin boxes: direct form, neighbours through L1/L2
    if (thin)
        hipLaunchKernelGGL(shallow_step_direct_cols, dim3((nyb + 255) / 256), dim3(256), 0, (hipStream_t)stream, *q, ld,
                           xstart - 1, xstop - 1, ystart - 1, ystop - 1, u, v, p, uold, vold, pold, unew, vnew, pnew);
    else
        hipLaunchKernelGGL(shallow_step_direct, dim3((nx + 255) / 256, nyb > 4096 ? 4096 : nyb), dim3(256), 0,
                           (hipStream_t)stream, *q, ld, xstart - 1, xstop - 1, ystart - 1, ystop - 1, u, v, p, uold,
                           vold, pold, unew, vnew, pnew);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

static dim3 grid2d(int nx, int ny)
{
    int gx = (nx + 255) / 256;
    if (gx > 64) gx = 64;
    int gy = ny > 4096 ? 4096 : ny;
    return dim3(gx, gy);
}

extern "C" int dlesm_copy_patch_f64(const double *src, double *dst, int ld, int ny_arr, int sx0, int sy0,
                                    int dx0, int dy0, int nx, int ny, void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(src && dst, "null pointer");
    if (nx <= 0 || ny <= 0) return DLESM_OK;
    if (int rc = check_box("dlesm_copy_patch_f64(src)", ld, ny_arr, sx0, sx0 + nx - 1, sy0, sy0 + ny - 1, 0)) return rc;
    if (int rc = check_box("dlesm_copy_patch_f64(dst)", ld, ny_arr, dx0, dx0 + nx - 1, dy0, dy0 + ny - 1, 0)) return rc;
    if (src == dst) {
        const bool overlap = !(dx0 + nx <= sx0 || sx0 + nx <= dx0 || dy0 + ny <= sy0 || sy0 + ny <= dy0);
        DLESM_REQUIRE(!overlap, "copy_patch: overlapping source and destination patches");
    }
    const size_t n = (size_t)nx * ny;
    const double *s0 = src + lin(ld, sx0, sy0);
    double *d0 = dst + lin(ld, dx0, dy0);
    if (nx == ld && n % 2 == 0 && (uintptr_t)s0 % 16 == 0 && (uintptr_t)d0 % 16 == 0 && n / 2 < ((size_t)1 << 31) * 256) {
        // whole rows (field_copy_code over a whole field, copy_field of a field): contiguous
        if (nt_stores_for(ld, 0, ny - 1))
            hipLaunchKernelGGL(copy_linear_k<true>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               s0, d0, n / 2);
        else
            hipLaunchKernelGGL(copy_linear_k<false>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               s0, d0, n / 2);
    } else {
        hipLaunchKernelGGL(copy_patch_k, grid2d(nx, ny), dim3(256), 0, (hipStream_t)stream, src, dst, ld,
                           sx0 - 1, sy0 - 1, dx0 - 1, dy0 - 1, nx, ny);
    }
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

extern "C" int dlesm_fill_f64(double *f, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                              double value, void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(f != nullptr, "null pointer");
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("dlesm_fill_f64", ld, ny, xstart, xstop, ystart, ystop, 0)) return rc;
    const int nx = xstop - xstart + 1, nyb = ystop - ystart + 1;
    int segs, segp;
    rowseg_split(nx, tuning("util_segp", SEG_PAIRS), &segs, &segp);
    const size_t n = (size_t)nx * nyb;
    double *f0 = f + lin(ld, xstart, ystart);
    if (nx == ld && n % 2 == 0 && (uintptr_t)f0 % 16 == 0 && n / 2 < ((size_t)1 << 31) * 256 && tuning("util_rowseg", 1))
        hipLaunchKernelGGL(fill_linear_k, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f0, n / 2, value,
                           nt_stores_for(ld, ystart - 1, ystop - 1) != 0);
    else if (rowlinear_ok(f, ld, nx, nyb) && tuning("util_rowseg", 1)) {
        const size_t n2 = (size_t)(ld / 2) * nyb;
        hipLaunchKernelGGL((rowlinear_write_k<GenConst>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, ld,
                           xstart - 1, ystart - 1, nx, n2, nt_stores_for(ld, ystart - 1, ystop - 1) != 0, GenConst{value});
    } else if ((uintptr_t)f % 16 == 0 && nx >= ROWSEG_MIN_NX && (long)segs * nyb < (1L << 31) && tuning("util_rowseg", 1))
        hipLaunchKernelGGL((rowseg_write_k<GenConst>), dim3((unsigned)((long)segs * nyb)), dim3(256), 0, (hipStream_t)stream, f, ld,
                           xstart - 1, ystart - 1, nx, segs, segp, nt_stores_for(ld, ystart - 1, ystop - 1) != 0, GenConst{value});
    else
        hipLaunchKernelGGL(fill_k, grid2d(nx, nyb), dim3(256), 0, (hipStream_t)stream, f, ld, xstart - 1,
                           ystart - 1, nx, nyb, value);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

extern "C" int dlesm_hash_init_f64(double *f, int ld, int ny, int xstart, int xstop, int ystart,
                                   int ystop, uint64_t seed, int64_t gx0, int64_t gy0, void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(f != nullptr, "null pointer");
    if (xstop < xstart || ystop < ystart) return DLESM_OK;
    if (int rc = check_box("dlesm_hash_init_f64", ld, ny, xstart, xstop, ystart, ystop, 0)) return rc;
    const int nx = xstop - xstart + 1, nyb = ystop - ystart + 1;
    int segs, segp;
    rowseg_split(nx, tuning("util_segp", SEG_PAIRS), &segs, &segp);
    if (rowlinear_ok(f, ld, nx, nyb) && tuning("util_rowseg", 1)) {
        const size_t n2 = (size_t)(ld / 2) * nyb;
        hipLaunchKernelGGL((rowlinear_write_k<GenHash>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, ld,
                           xstart - 1, ystart - 1, nx, n2, nt_stores_for(ld, ystart - 1, ystop - 1) != 0, GenHash{seed, gx0, gy0});
    } else if ((uintptr_t)f % 16 == 0 && nx >= ROWSEG_MIN_NX && (long)segs * nyb < (1L << 31) && tuning("util_rowseg", 1))
        hipLaunchKernelGGL((rowseg_write_k<GenHash>), dim3((unsigned)((long)segs * nyb)), dim3(256), 0, (hipStream_t)stream, f, ld,
                           xstart - 1, ystart - 1, nx, segs, segp, nt_stores_for(ld, ystart - 1, ystop - 1) != 0,
                           GenHash{seed, gx0, gy0});
    else
        hipLaunchKernelGGL(hash_init_k, grid2d(nx, nyb), dim3(256), 0, (hipStream_t)stream, f, ld, xstart - 1,
                           ystart - 1, nx, nyb, seed, gx0, gy0);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// SUM(ABS(box)) enqueued on `s`: partials (one per row segment) -> levels of 4096 -> *result_dev.  `scratch` holds
// segs*rows + (segs*rows + 4095)/4096 doubles.
static int enqueue_checksum(const double *f, int ld, int x0, int y0, int nx, int nyb, double *scratch, double *result_dev,
                            hipStream_t s)
{
    if (nx < ROWSEG_MIN_NX) {                                // a few columns: the whole-row form
        const int blocks = nyb < kMaxPartials ? nyb : kMaxPartials;
        hipLaunchKernelGGL(abs_sum_rows, dim3(blocks), dim3(256), 0, s, f, ld, x0, y0, nx, nyb, scratch);
        hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, scratch, blocks, result_dev);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    int segs, segp;
    rowseg_split(nx, tuning("util_segp", SEG_PAIRS), &segs, &segp);
    const long n1 = (long)segs * nyb;
    hipLaunchKernelGGL(abs_sum_rowseg, dim3((unsigned)n1), dim3(256), 0, s, f, ld, x0, y0, nx, segs, segp, scratch);
    const double *lvl = scratch;
    long n = n1;
    if (n > 4096) {
        const long n2 = (n + 4095) / 4096;                       // <= 2^31 / 4096 row segments: fits one more level
        hipLaunchKernelGGL(sum_partials_level, dim3((unsigned)n2), dim3(256), 0, s, lvl, n, 4096, scratch + n1);
        lvl = scratch + n1;
        n = n2;
    }
    DLESM_REQUIRE(n <= (1 << 20), "checksum: box too large");
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, lvl, (int)n, result_dev);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}
static long checksum_scratch_doubles(int nx, int nyb)
{
    int segs, segp;
    rowseg_split(nx, tuning("util_segp", SEG_PAIRS), &segs, &segp);
    const long n1 = (long)segs * nyb;
    const long need = n1 + (n1 + 4095) / 4096 + 1;
    return need > kMaxPartials + 1 ? need : kMaxPartials + 1;
}

static double *g_cs_scratch = nullptr;   // grows with the largest box seen (g_scratch_mu)
static long g_cs_scratch_n = 0;

extern "C" int dlesm_checksum_f64(const double *f, int ld, int ny, int xstart, int xstop, int ystart,
                                  int ystop, double *result, void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(f != nullptr && result != nullptr, "null pointer");
    if (xstop < xstart || ystop < ystart) {
        *result = 0.0;
        return DLESM_OK;
    }
    if (int rc = check_box("dlesm_checksum_f64", ld, ny, xstart, xstop, ystart, ystop, 0)) return rc;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    const int nx = xstop - xstart + 1, nyb = ystop - ystart + 1;
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)f % 16 == 0 && tuning("util_rowseg", 1)) {
        const long need = checksum_scratch_doubles(nx, nyb);
        if (need > g_cs_scratch_n) {
            if (g_cs_scratch) DLESM_HIP_TRY(hipFree(g_cs_scratch));      // (synchronises: nothing of ours is in flight under the lock)
            g_cs_scratch = nullptr, g_cs_scratch_n = 0;
            DLESM_HIP_TRY(hipMalloc((void **)&g_cs_scratch, (size_t)need * sizeof(double)));
            g_cs_scratch_n = need;
        }
        if (int rc = enqueue_checksum(f, ld, xstart - 1, ystart - 1, nx, nyb, g_cs_scratch, g_cs_scratch + need - 1, s)) return rc;
        DLESM_HIP_TRY(hipMemcpyAsync(result, g_cs_scratch + need - 1, sizeof(double), hipMemcpyDeviceToHost, s));
        DLESM_HIP_TRY(hipStreamSynchronize(s));
        return DLESM_OK;
    }
    if (!g_partials) DLESM_HIP_TRY(hipMalloc((void **)&g_partials, (kMaxPartials + 1) * sizeof(double)));
    const int blocks = nyb < kMaxPartials ? nyb : kMaxPartials;
    hipLaunchKernelGGL(abs_sum_rows, dim3(blocks), dim3(256), 0, s, f, ld, xstart - 1, ystart - 1, nx, nyb,
                       g_partials);
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, g_partials, blocks, g_partials + kMaxPartials);
    DLESM_HIP_TRY(hipGetLastError());
    DLESM_HIP_TRY(hipMemcpyAsync(result, g_partials + kMaxPartials, sizeof(double), hipMemcpyDeviceToHost, s));
    DLESM_HIP_TRY(hipStreamSynchronize(s));
    return DLESM_OK;
}

// The same sum WITHOUT a host synchronisation: *result_dev (device memory, or host memory the device can write:
// hipHostMalloc) receives the value when `stream` gets there.  What a time loop that logs a checksum every few steps
// wants: no pipeline drain per check; read the values after the loop's own synchronisation.  Deterministic, and equal
// to dlesm_checksum_f64 bit for bit.  The scratch space is stream-ordered (hipMallocAsync / hipFreeAsync).
extern "C" int dlesm_checksum_async_f64(const double *f, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                                        double *result_dev, void *stream)
{
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(f != nullptr && result_dev != nullptr, "null pointer");
    DLESM_REQUIRE((uintptr_t)f % 16 == 0, "dlesm_checksum_async_f64: field not 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (xstop < xstart || ystop < ystart) {
        DLESM_HIP_TRY(hipMemsetAsync(result_dev, 0, sizeof(double), s));
        return DLESM_OK;
    }
    if (int rc = check_box("dlesm_checksum_async_f64", ld, ny, xstart, xstop, ystart, ystop, 0)) return rc;
    const int nx = xstop - xstart + 1, nyb = ystop - ystart + 1;
    double *scratch = nullptr;
    DLESM_HIP_TRY(hipMallocAsync((void **)&scratch, (size_t)checksum_scratch_doubles(nx, nyb) * sizeof(double), s));
    const int rc = enqueue_checksum(f, ld, xstart - 1, ystart - 1, nx, nyb, scratch, result_dev, s);
    DLESM_HIP_TRY(hipFreeAsync(scratch, s));
    return rc;
}

extern "C" int dlesm_probe_stream_concurrency(void *stream)
{
    if (int rc = ensure_device()) return rc;
    {
        std::lock_guard<std::mutex> lk(g_probe_mu);
        g_probe_seen.erase((hipStream_t)stream);
    }
    return streams_run_concurrently((hipStream_t)stream) ? 1 : 0;
}
