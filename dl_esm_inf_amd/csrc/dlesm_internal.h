// Internal helpers shared by the translation units of libdlesm_hip.so.
#ifndef DLESM_INTERNAL_H
#define DLESM_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "dlesm_error.h"
#include "dlesm_hip.h"

namespace dlesm {

#define DLESM_HIP_TRY(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::dlesm::fail(DLESM_EHIP, "%s failed: %s (%s:%d)", #expr,                \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

#define DLESM_REQUIRE(cond, ...)                                                            \
    do {                                                                                    \
        if (!(cond)) return ::dlesm::fail(DLESM_EINVAL, __VA_ARGS__);                       \
    } while (0)

// true once a device has been bound (dlesm_init, or lazily on first device call)
int ensure_device();
hipStream_t side_stream();     // created by ensure_device()
hipStream_t transfer_stream(); // for the non-blocking sync callbacks
int tuning(const char *key, int fallback);
// DLESM_LAB: the measurement build of the SAME sources (libdlesm_hip_lab.so): comparison-only kernels (the y-march and
// LDS-staged Jacobi sweeps, taller / shorter wave tiles, shuffle instead of DPP, stacked and straight-line shallow-water
// tiles, the pipeline form of the fused steps) and diagnostics that skip work are compiled only there; the product library
// holds the forms it can reach by itself.  A LAB tuning key reads as its fallback in the product (dlesm_runtime.hip).
#ifdef DLESM_LAB
constexpr bool kLab = true;
#else
constexpr bool kLab = false;
#endif
// boxes of the shallow-water sweeps this many columns wide or narrower (the west / east frame columns of a distributed step)
// take the one-cell-per-thread form: a wave tile would load 128 columns for them
constexpr int SW_THIN_BOX = 8;
// the process-wide pinned word a device-side wait raises when it gives up (checked by every device entry point)
int *wait_timed_out_word();
// forget what streams_run_concurrently has measured (a time-out was acknowledged, the runtime was finalised)
void invalidate_concurrency_probe();
// bound, in ticks of the 100 MHz counter, of a wait whose release depends on OTHER ranks (an exchange completing):
// dm_wait_seconds, default 600; 0 = no limit, as the reference waits in MPI_Waitany
unsigned long long remote_wait_ticks();

// column-major 1-based -> linear offset (field_mod.f90:350)
__host__ __device__ inline size_t lin(int ld, int ji, int jj)
{
    return (size_t)(jj - 1) * (size_t)ld + (size_t)(ji - 1);
}

// Whole-wave shifts by one lane on the VALU (DPP wave_shr:1 / wave_shl:1 of the GFX9 family)
// instead of ds_bpermute through the LDS pipe: from_lower(x) is lane-1's x, from_upper(x) is
// lane+1's x, without occupying the one LDS unit the 4 SIMDs of a CU share.  The lane that has
// no source (lane 0 / lane 63) gets 0.0 with DPP (bound_ctrl, no copy of the old value needed)
// and its own x with the shuffle: callers must not use that lane's result.
template <bool DPP>
__device__ __forceinline__ double from_lower(double x)
{
    if constexpr (DPP) {
        const int lo = __double2loint(x), hi = __double2hiint(x);
        return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true),
                                __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true));
    } else {
        return __shfl_up(x, 1);
    }
}
template <bool DPP>
__device__ __forceinline__ double from_upper(double x)
{
    if constexpr (DPP) {
        const int lo = __double2loint(x), hi = __double2hiint(x);
        return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true),
                                __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true));
    } else {
        return __shfl_down(x, 1);
    }
}

// The value unchanged, through a CONVERGENT operation (a DPP move with the identity lane pattern quad_perm:[0,1,2,3]).
// Code motion may not make a convergent operation control-dependent on a lane-varying condition, so whatever feeds
// pin_here() -- in particular the LOADS behind it -- cannot be sunk into the masked store branches of a kernel: without
// it the compiler moves a load whose only use is a conditional store into that branch, and a lane's loads are issued
// one dependent round trip after the other instead of all together (seen in the ISA of the gather copies and of the
// old-level loads of the shallow-water step).
__device__ __forceinline__ double pin_here(double x)
{
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0xE4, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(x), 0xE4, 0xf, 0xf, true));
}

// shared by the frame/interior split of the distributed step (dlesm_halo.hip)
int launch_stencil5(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                    int ystart, int ystop, hipStream_t s);
// columns of the frame that are also send strips: the frame kernel writes their cells into
// the halo plan's send buffer as it computes them (no separate pack launch)
struct FramePack {
    static constexpr int MAXS = 4;
    struct Col { int i, j0, nj; long off; } s[MAXS]; // 0-based column, first row, rows, slot offset
    int n;
    double *buf;
};
int launch_stencil5_frame(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                          int ystart, int ystop, hipStream_t s, const FramePack *pack = nullptr);
// non-temporal stores for a sweep over rows y0..y1 of arrays of leading dimension ld? (dlesm_kernels.hip)
int nt_stores_for(int ld, int y0, int y1);

// general 3x3 weighted stencil (dlesm_stencil9.hip): the sweep over a box, and the one-cell frame of a
// box in one launch (its west/east columns also written into the send buffer)
int launch_stencil9(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                    int ystart, int ystop, hipStream_t s);
int launch_stencil9_frame(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                          int ystart, int ystop, hipStream_t s, const FramePack *pack);

// (declared after FrameJob, below) launch_stencil9_framed

// the one-cell frame of the shallow-water step in ONE launch (one cell per thread).  Every frame cell
// that a neighbour will receive (rows, columns and corners) is also written into the AGGREGATED send
// buffer of the three new fields: one message per neighbour and direction carries all three strips,
// field after field, each in the pack loop's order (j outer, i inner; parallel_comms_mod.f90:1678-1683).
struct FramePack3 {
    static constexpr int MAXS = 8;
    struct S { int i0, j0, ni, nj; long off; } s[MAXS];   // 0-based strip; off = offset of the message's slot
    int n;
    double *buf;                  // aggregated send buffer; field k of strip q at off + k*ni*nj
    double *base[MAXS];           // peer transport: strip q goes to base[q] + slot (off = 0) -- a neighbour's mailbox; null: buf
    __host__ __device__ double *at(int q) const { return base[q] ? base[q] : buf; }
    __host__ __device__ long slot(int q, int k, int i, int j) const
    {
        return s[q].off + (long)k * s[q].ni * s[q].nj + (long)(j - s[q].j0) * s[q].ni + (i - s[q].i0);
    }
    __host__ __device__ bool holds(int q, int i, int j) const
    {
        return i >= s[q].i0 && i < s[q].i0 + s[q].ni && j >= s[q].j0 && j < s[q].j0 + s[q].nj;
    }
};
int launch_shallow_frame(const dlesm_sw_params &q, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         const double *u, const double *v, const double *p, const double *uold,
                         const double *vold, const double *pold, double *unew, double *vnew, double *pnew,
                         const FramePack3 *pack, hipStream_t s, const double *smooth_alpha = nullptr);

// the frame as the first workgroups of the interior launch (jacobi5_tile_framed): when their last
// one is done, `seq` is stored to `flag` (device memory; frame_flag_wait sleeps on it)
struct FrameJob {
    FramePack pk;
    int fx0, fx1, fy0, fy1;       // 0-based frame box (filled in by launch_stencil5_framed)
    int nblocks;                  // workgroups that do frame cells (filled in by the launcher)
    unsigned *counter;            // device word, 0 between launches
    unsigned long long *flag;
    unsigned long long seq;
    // pipelined steps: before touching anything, the frame workgroups wait (bounded) until
    // *halo_flag >= halo_seq -- the previous step's exchange has landed in `in`'s halos
    const unsigned long long *halo_flag;
    unsigned long long halo_seq;  // 0: no wait
    unsigned long long halo_wait_ticks;   // bound of that wait (remote_wait_ticks(); 0 = none)
    int *timed_out;               // pinned host word raised by a wait that gives up
    int acquire;                  // dm_acquire (default 1): one agent-scope acquire behind that wait (handover_acquire)
    // pipelined steps: the west/east halo columns of `in` have NOT been unpacked into the field; they
    // are read from the receive buffer of the previous exchange (contiguous), strip by strip
    struct HaloCol { int i, j0, nj; long off; } hs[FramePack::MAXS]; // 0-based halo column, first row, rows, slot
    int nh;
    const double *halo_buf;       // nullptr: halos are in the field
};
// Frame of the box + interior sweep in ONE launch.  *fused = false (and nothing launched) when the
// arrays do not qualify for the 16-byte-lane tile kernel: the caller then takes the two-launch path.
int launch_stencil5_framed(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart,
                           int ystop, FrameJob job, hipStream_t s, bool *fused);
// the same for the 3x3 weighted stencil (joined form only: no halo wait, no virtual halos)
int launch_stencil9_framed(const double *in, double *out, const double *coef, int ld, int ny, int xstart, int xstop,
                           int ystart, int ystop, FrameJob job, hipStream_t s, bool *fused);
// park stream `s` (one sleeping wave) until *flag >= seq; bounded, see frame_flag_wait
// remote = the flag is raised when an EXCHANGE has completed (depends on other ranks): bounded by remote_wait_ticks()
// instead of the 30 s that bound a wait for this GPU's own frame workgroups
int launch_frame_flag_wait(const unsigned long long *flag, unsigned long long seq, int *timed_out, hipStream_t s,
                           bool remote = false);
// true when kernels of two streams execute side by side in this process (probed once; false under
// kernel-serialising tools): precondition of the one-launch / time-loop forms of the distributed step
bool streams_run_concurrently(hipStream_t callers);
// *flag = seq, stream ordered (one thread): publishes "the exchange before this point has landed"
int launch_flag_set(unsigned long long *flag, unsigned long long seq, hipStream_t s);

// ---- peer transport (dlesm_halo_plan_peer_connect): no RCCL kernel, no side stream.  The frame workgroups of the step
// launch store every cell a neighbour needs STRAIGHT INTO THAT NEIGHBOUR'S receive mailbox (peer-mapped memory: xGMI
// stores) and then raise the neighbour's arrival flag; they read their own halo operands from the local mailbox once its
// arrival flags are up.  Mailboxes are double-buffered on the step's sequence number.
// The sequence number of a mailbox operation LIVES ON THE DEVICE, so that operations captured into a hipGraph advance from
// replay to replay: three words per plan, w[0] / w[1] = the number of the next operation of even / odd parity, w[2] = the number of
// the last operation whose flags have been raised.  Every kernel of an operation the host counts as `host_seq` (whose parity
// selected the mailbox half on the host) takes its number from w[host_seq & 1]; the ONE workgroup of the operation that raises the
// neighbours' flags then stores seq + 1 into the OTHER word -- the next operation's, which no workgroup of this operation reads,
// and which the next operation (behind this one in stream order) reads only after this launch has drained.  In a process that
// never replays a graph the device number equals the host's.  A graph must hold an EVEN number of mailbox operations of a plan
// (the parities are baked into its pointers); one replayed out of step finds w[2] != seq - 1 and raises the sticky word (value 2).
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long peer_seq_load(const unsigned long long *w, unsigned long long host_seq)
{
    return w ? __hip_atomic_load(w + (host_seq & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : host_seq;
}
__device__ __forceinline__ void peer_seq_advance(unsigned long long *w, unsigned long long seq, int *sticky)
{
    if (!w) return;
    if (__hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq - 1 && sticky)
        __hip_atomic_store(sticky, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 2, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + ((seq + 1) & 1), seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
struct PeerJob {
    static constexpr int MAXM = 8;
    int fx0, fx1, fy0, fy1;       // 0-based frame box (filled in by launch_stencil5_peer)
    int nblocks;                  // workgroups that do frame cells (filled in by the launcher)
    unsigned *counter;            // device word, 0 between launches
    // halo operands of this step: strips of the LOCAL mailbox (parity of the previous step), each with its arrival flag
    struct In { int i0, j0, ni, nj; const double *src; const unsigned long long *flag; } in[MAXM];
    int nin;
    unsigned long long wait_seq;  // the frame workgroups wait until every in[k].flag >= wait_seq; 0: no wait
    int virt;                     // != 0: halo operands are read from the strips; 0: from the field (already unpacked)
    // what the neighbours get: strips of the frame, each into a PEER's mailbox (parity of this step) + that peer's flag
    struct Out { int i0, j0, ni, nj; double *dst; unsigned long long *flag; } out[MAXM];
    int nout;
    unsigned long long seq;       // the operation's number as the HOST counts it (its parity picked the mailbox halves above); what the
                                  // last frame workgroup stores to every out[k].flag is the device's number, peer_seq_load(seqw, seq)
    unsigned long long *seqw;     // the plan's sequence words on the device (peer_seq_load): the number the kernels USE
    unsigned long long wait_ticks;   // bound of the wait (remote_wait_ticks(); 0 = none)
    int *timed_out;
    int fenced;                   // mailbox_fences: release store of the flags, acquire fence behind the wait (peer_raise_flag)
    // joined form in ONE launch: `nunb` workgroups behind the frame workgroups wait for THIS step's arrival flags and copy the
    // received strips (un[], parity of this step) into the halo cells of `out` -- the join without a second launch.  They
    // depend on the NEIGHBOURS' frame workgroups only (dispatched first in their launches, as ours are in this one).
    In un[MAXM];
    int nun, nunb;                // strips, workgroups (a multiple of 8, so that the tile workgroups keep their XCD); 0: none
};
// frame (peer stores) + interior sweep in ONE launch; *fused = false (nothing launched) when the arrays do not qualify
int launch_stencil5_peer(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         PeerJob job, hipStream_t s, bool *fused);
// the frame workgroups alone (any alignment, boxes without an interior); the caller sweeps the interior itself
int launch_stencil5_peer_frame(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                               PeerJob job, hipStream_t s);
// wait for the arrival flags of n strips (>= seq), then copy them from the mailbox into the halo cells of `field`
struct PeerStrips { PeerJob::In s[PeerJob::MAXM]; int n; };
// a halo exchange over the mailboxes, first half: every enabled send strip of nf fields copied into the neighbours'
// mailboxes (field after field inside a message), then -- by the last workgroup to finish -- `seq` into their arrival flags
struct PeerOuts { PeerJob::Out s[PeerJob::MAXM]; int n; };
int launch_peer_pack(const PeerOuts &out, const double *const *fields, int nf, int ld, unsigned *counter, unsigned long long seq,
                     unsigned long long *seqw, int *sticky, hipStream_t s);
// both halves of an exchange in one launch (pack half dispatched first; the unpack half waits for the neighbours only)
struct PeerStrips;
int launch_peer_exchange(const PeerOuts &out, const PeerStrips &in, double *const *fields, int nf, int ld, unsigned *counter,
                         unsigned long long seq, unsigned long long *seqw, int *timed_out, hipStream_t s);
// nf fields: field k of a strip sits k*ni*nj doubles behind its first (the aggregated layout)
int launch_peer_unpack(const PeerStrips &st, unsigned long long seq, const unsigned long long *seqw, double *const *fields, int nf, int ld,
                       int *timed_out, hipStream_t s);
// *flag[k] = seq for n flags in peer memory (system scope), stream ordered: behind a frame launch that is not the fused one
int launch_peer_flags_set(unsigned long long *const *flags, int n, unsigned long long seq, unsigned long long *seqw, int *sticky,
                          hipStream_t s);

// nsteps fused Jacobi steps (dlesm_jacobi_x2.hip); 1-based inclusive output box, last stage box,
// grow flags -- see dlesm_stencil5_multi_f64
int launch_stencil5_multi(const double *in, double *out, int ld, int ny, int nsteps, int xstart, int xstop,
                          int ystart, int ystop, int exstart, int exstop, int eystart, int eystop, int gw,
                          int ge, int gs, int gn, hipStream_t s);

// block shape (waves per workgroup, padded tiles per row) of a linear tile sweep
void choose_block_shape(int *nxw_io, int *tpb_out, int prefer = 0);
// the same for a sweep with the Jacobi tile geometry over the 0-based box: the planned Jacobi shape of that (ld, box) if there is one
void shape_for_tile_sweep(int ld, int x0, int x1, int y0, int y1, int *nxw_io, int *tpb_out);
int check_box(const char *who, int ld, int ny, int xstart, int xstop, int ystart, int ystop, int ring);
// the shallow-water frame as the first workgroups of the interior launch (shallow_tile_framed)
struct SwFrameJob {
    FramePack3 pk;
    int fx0, fx1, fy0, fy1;       // 0-based frame box
    int nblocks;                  // filled in by the launcher
    unsigned *counter;
    unsigned long long *flag;
    unsigned long long seq;
    // pipelined steps (as FrameJob): the frame workgroups first wait until *halo_flag >= halo_seq -- the
    // previous step's exchange has landed in the halos of u, v, p and is done with the send buffers
    const unsigned long long *halo_flag;
    unsigned long long halo_seq;  // 0: no wait
    unsigned long long halo_wait_ticks;   // bound of that wait (remote_wait_ticks(); 0 = none)
    int *timed_out;
    int acquire;                  // dm_acquire, as FrameJob
    int smooth;                   // != 0: the Asselin filter of the old level folded in (time_smooth, coefficient alpha)
    double alpha;
    // peer transport: the pack strips are the neighbours' mailboxes (system-scope stores) and the last frame workgroup
    // raises THEIR arrival flags with `seq` instead of the local frame flag
    int npeer;
    unsigned long long *peer_flag[FramePack3::MAXS];
    int fenced;                   // mailbox_fences (see PeerJob)
    unsigned long long *seqw;     // peer transport: the plan's sequence words (peer_seq_load); nullptr: `seq` as passed
    // ... and the join inside the launch (as PeerJob::un): `nunb` workgroups behind the ring workgroups wait for THIS step's
    // arrival flags and copy the received strips -- three fields per message -- into the halos of unew, vnew, pnew
    PeerJob::In un[PeerJob::MAXM];
    int nun, nunb;
    int diag;                     // profiling only (results wrong): 1 = no frame cells, 2 = south/north rows only
};
// shallow-water step, register-tiled linear sweep (dlesm_shallow.hip); 0-based inclusive box
void launch_shallow_tile(const dlesm_sw_params &q, int ld, int x0, int x1, int y0, int y1,
                         const double *u, const double *v, const double *p, const double *uold,
                         const double *vold, const double *pold, double *unew, double *vnew,
                         double *pnew, hipStream_t s, bool sw_offset = false, SwFrameJob *fj = nullptr, int wrap = 0,
                         const double *smooth_alpha = nullptr);   // non-null: also uold/vold/pold <- time_smooth, in place
// frame of the box + interior sweep of the NE shallow-water step in one launch; *fused = false (nothing
// launched) when the arrays do not qualify for the tile kernel
int launch_shallow_framed(const dlesm_sw_params &q, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                          const double *u, const double *v, const double *p, const double *uold,
                          const double *vold, const double *pold, double *unew, double *vnew, double *pnew,
                          SwFrameJob job, hipStream_t s, bool *fused, const double *smooth_alpha = nullptr);

} // namespace dlesm

struct dlesm_field {
    uint64_t magic;
    double *data;
    int ld, ny;
    bool owned;
};
static const uint64_t DLESM_FIELD_MAGIC = 0x444c45534d464c44ULL; // "DLESMFLD"

#endif
