// Device-resident halo exchange: the MI355X replacement for
// r2d_field%halo_exchange -> exchange_generic -> MPI_Irecv/Isend/Waitany
// (field_mod.f90:1231-1256, parallel_comms_mod.f90:1501-1855,
//  parallel/parallel_utils_mod.f90:148-211).
//
// Design (not a translation of the MPI call pattern):
//   * fields never leave HBM; the D2H/H2D strip copies of field_mod.f90:1241-1254
//     disappear;
//   * y-strips and corner cells are contiguous in memory (nysend = 1), so they are
//     sent from / received into the field itself; only the strided x-strips are
//     packed, by ONE kernel launch for all of them, into a persistent buffer;
//   * all messages of an exchange go out in a single ncclGroupStart/End: every
//     mesh neighbour of a rank is a direct xGMI peer, messages are <= 128 KiB, so
//     the exchange is latency bound and what matters is one launch, not link
//     bandwidth;
//   * everything is stream ordered: the waits of msg_wait/msg_wait_all become
//     stream dependencies, and dlesm_jacobi5_step_dm hides the whole exchange
//     behind the interior stencil on a side stream.
//
// Message matching: RCCL has no tags.  Between a given pair of ranks messages
// match in issue order, so sends to and receives from one peer are both issued
// in ascending direction code -- the code is the MPI tag offset of the
// reference (tag_orig + dir, pcomms:1606,1647) and is the same number on the
// sending and on the receiving side of a message.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/file.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <vector>

#include "dlesm_internal.h"
#include "dlesm_device.h"

using namespace dlesm;

#define DLESM_NCCL_TRY(expr)                                                                \
    do {                                                                                    \
        ncclResult_t _r = (expr);                                                           \
        if (_r != ncclSuccess)                                                              \
            return ::dlesm::fail(DLESM_ERCCL, "%s failed: %s (%s:%d)", #expr,               \
                                 ncclGetErrorString(_r), __FILE__, __LINE__);               \
    } while (0)

// Inside an open ncclGroupStart/ncclGroupEnd pair a failed call must not return at once: the
// thread's group would stay open and swallow every later RCCL call.  Remember the first error,
// skip the rest, and report after ncclGroupEnd.
#define DLESM_NCCL_IN_GROUP(first_err, expr)                                                \
    do {                                                                                    \
        if ((first_err) == ncclSuccess) (first_err) = (expr);                               \
    } while (0)

static_assert(sizeof(ncclUniqueId) == DLESM_UNIQUE_ID_BYTES, "ncclUniqueId size");

namespace {

ncclComm_t g_comm = nullptr;
int g_rank = -1, g_size = 0;
// MAILBOX MODE (dlesm_comm_init_mailbox): no RCCL communicator at all.  Every plan connects its mailboxes when it is
// created (the descriptors travel over the host-side board, dlesm_rendezvous.cpp), exchanges and distributed steps go
// through them, a global sum is eight bytes over the board, a gather is every rank copying its block straight into the
// root's buffer through an IPC mapping.  What a job of one process per GPU needs from MPI in the reference, without a
// communication library underneath.
bool g_mailbox = false;

struct Strip { // one packed (strided) message
    int i0, j0, nx, ny; // 0-based origin and extent inside the field
    long off;           // offset (doubles) of its slot in the pack buffer
    int dir;            // direction code, for masked exchanges
};

int group_end(ncclResult_t first_err, const char *what)
{
    const ncclResult_t end = ncclGroupEnd();
    if (first_err != ncclSuccess)
        return fail(DLESM_ERCCL, "%s: %s", what, ncclGetErrorString(first_err));
    if (end != ncclSuccess) return fail(DLESM_ERCCL, "%s: ncclGroupEnd: %s", what, ncclGetErrorString(end));
    return DLESM_OK;
}

struct Msg {
    int dir, peer;
    int i0, j0, nx, ny; // 0-based
    long count;
    long off;           // pack-buffer slot, or -1 when sent/received in place
};

} // namespace

struct dlesm_halo_plan {
    int ld, ny;
    std::vector<Msg> sends, recvs; // sorted by (peer, dir)
    int n_spack = 0, n_rpack = 0;
    Strip *d_spack = nullptr, *d_rpack = nullptr; // device tables
    double *sendbuf = nullptr, *recvbuf = nullptr;
    long sendbuf_len = 0, recvbuf_len = 0;   // doubles per field
    int buf_fields = 0;                      // fields the buffers currently have room for
    int max_strip = 0;
    // Aggregated exchanges of several fields (dlesm_halo_exchange_multi_f64, the shallow-water step): ONE
    // message per neighbour and direction carries the strips of all nf fields, field after field --
    // measured in loop-back, an RCCL group costs ~4 us per send/recv pair whatever its size up to a
    // row of 8192 doubles, so 3 fields x 8 directions as 24 messages take 3x as long as 8.  Every
    // message (rows too) then has a slot: message m of nf fields lives at nf * agg[m], count * nf doubles.
    std::vector<long> sagg, ragg;            // per message, in the sorted order of sends / recvs
    long sagg_len = 0, ragg_len = 0;         // doubles per field
    Strip *d_sall = nullptr, *d_rall = nullptr;   // device tables of ALL messages (off = agg offset)
    int max_msg = 0;
    double *sendagg = nullptr, *recvagg = nullptr;
    int agg_fields = 0;
    // frame-done / exchange-done events of the overlapped steps: per plan, so that two grids
    // stepped on different streams never share one.  ONE exchange per plan may be in flight
    // (its pack buffers are single): the steps below serialise on the caller's stream.
    hipEvent_t ev_frame = nullptr, ev_comm = nullptr;
    // frame-done flag of the one-launch step (jacobi5_tile_framed): a device word the side stream's
    // frame_flag_wait kernel sleeps on, a device word counting finished frame workgroups, the value
    // the next launch will store (monotonic, so "flag >= seq" can never be missed), and a pinned
    // host word the wait kernel raises if it ever gives up
    unsigned long long *frame_flag = nullptr;
    unsigned *frame_counter = nullptr;
    unsigned long long frame_seq = 0;
    int *frame_timed_out = nullptr;
    // pipelined steps (dlesm_jacobi5_step_dm_pipelined): the exchange of the last step is still in
    // flight on the side stream when the call returns; `halo_flag` (device word, set on the side
    // stream after the unpack) is what the NEXT step's frame workgroups wait on, ev_comm what any
    // other consumer is joined on (dlesm_halo_plan_join)
    unsigned long long *halo_flag = nullptr;
    bool pending = false;              // an exchange has been issued and not yet joined on pending_stream
    unsigned long long pending_seq = 0;
    hipStream_t pending_stream = nullptr;
    // ... and its west/east strips have not been unpacked: they sit in recvbuf, where the next
    // pipelined step's frame workgroups read them; whoever joins instead unpacks them into this field
    double *pending_field = nullptr;
    unsigned pending_mask = 0;
    // ---- peer transport (dlesm_halo_plan_peer_export / _connect) -------------------------
    // This rank's MAILBOX: one fine-grained allocation the neighbours store into over xGMI --
    //   [0, 256)      arrival flags, one 8-byte word per receive message (index = position in `recvs`)
    //   [4096, ...)   two payload parities of peer_par_len doubles: message m of an nf-field step at nf * ragg[m]
    // and, per SEND message, where its strip goes in the neighbour's mailbox (peer-mapped addresses).
    int peer_fcap = 0;                         // fields a payload parity has room for (0: no mailbox)
    void *peer_box = nullptr;
    unsigned long long *peer_flags = nullptr;
    double *peer_rx = nullptr;
    long peer_par_len = 0;
    bool peer_on = false;                      // connected
    std::vector<double *> peer_tx;             // per send: the neighbour's payload parity 0
    std::vector<long> peer_tx_par, peer_tx_off;   // its parity length; per-field offset of the matching slot
    std::vector<unsigned long long *> peer_txflag;
    std::vector<void *> peer_mapped;           // hipIpcOpenMemHandle results, closed with the plan
    unsigned *peer_counter = nullptr;
    unsigned long long *peer_seqw = nullptr;   // three device words behind the counter: the sequence numbers the kernels use (peer_seq_load)
    bool peer_pending_captured = false;        // the pending step was issued into a stream capture (its join may be captured too)
    unsigned long long peer_seq = 0;           // mailbox operations ISSUED by the host (the same number on every rank): its parity picks the mailbox half; the kernels take the number itself from peer_seqw
    bool peer_pending = false;                 // halos of pending_field are still in the mailbox (parity peer_seq & 1)
    hipEvent_t ev_peer = nullptr;
    // the mailboxes, their counter and the sequence number are ONE resource: an operation issued on another stream than
    // the previous one is ordered behind it (peer_order)
    bool peer_used = false;
    hipStream_t peer_last_stream = nullptr;
};

// edge directions follow their bit; diagonals follow their two edges (parallel_comms_mod.f90:
// 1557-1571) unless DLESM_DIRS_NO_DIAGONALS is set.  mask 0 enables nothing, as there.
__host__ __device__ static inline bool dir_enabled(unsigned mask, int dir)
{
    auto on = [&](int d) { return (mask >> (d - 1)) & 1u; };
    if (dir >= 1 && dir <= 4) return on(dir);
    if (mask & DLESM_DIRS_NO_DIAGONALS) return false;
    switch (dir) {
    case DLESM_IPLUSJPLUS: return on(DLESM_IPLUS) && on(DLESM_JPLUS);
    case DLESM_IMINUSJMINUS: return on(DLESM_IMINUS) && on(DLESM_JMINUS);
    case DLESM_IPLUSJMINUS: return on(DLESM_IPLUS) && on(DLESM_JMINUS);
    case DLESM_IMINUSJPLUS: return on(DLESM_IMINUS) && on(DLESM_JPLUS);
    default: return false;
    }
}

// ---- the host-only part of a plan: which messages, in which order, through which buffer slots ------------------
// Everything RCCL is told comes from here, and dlesm_halo_plan_describe reports exactly this, so that the ordering
// rule -- RCCL has no tags: between a pair of ranks the k-th send meets the k-th receive -- can be exercised by a
// tag-free transport on the CPU (tests/gloo_worker.py) against the product's OWN lists, not a re-statement of them.
struct MsgLists {
    std::vector<Msg> sends, recvs;           // sorted by (peer, dir)
    std::vector<Strip> spack, rpack;         // the strided strips of the single-field path, in list order
    long sendbuf_len = 0, recvbuf_len = 0;   // doubles per field
    int max_strip = 0;
    std::vector<long> sagg, ragg;            // aggregated path: slot offset per message (doubles per field)
    std::vector<Strip> sall, rall;
    long sagg_len = 0, ragg_len = 0;
    int max_msg = 0;
};

static bool by_peer_dir(const Msg &a, const Msg &b)
{
    return a.peer != b.peer ? a.peer < b.peer : a.dir < b.dir;
}

static int build_msg_lists(const dlesm_comm_tables *t, int ld, int ny, MsgLists &L)
{
    DLESM_REQUIRE(t != nullptr, "null pointer");
    DLESM_REQUIRE(ld > 0 && ny > 0, "field extents %dx%d", ld, ny);
    DLESM_REQUIRE(t->nsend >= 0 && t->nsend <= DLESM_MAXCOMM && t->nrecv >= 0 && t->nrecv <= DLESM_MAXCOMM,
                  "message counts %d/%d", t->nsend, t->nrecv);
    auto add = [&](std::vector<Msg> &list, int dir, int peer, int i1, int j1, int nx, int nyy) -> int {
        if (peer < 0 || nx <= 0) return DLESM_OK; // skipped by the reference too (pcomms:1603,1639)
        if (i1 < 1 || j1 < 1 || i1 + nx - 1 > ld || j1 + nyy - 1 > ny || nyy < 1)
            return fail(DLESM_EINVAL, "message patch (%d,%d)+%dx%d outside field %dx%d", i1, j1, nx, nyy, ld, ny);
        list.push_back(Msg{dir, peer, i1 - 1, j1 - 1, nx, nyy, (long)nx * nyy, -1});
        return DLESM_OK;
    };
    for (int k = 0; k < t->nsend; k++)
        if (int rc = add(L.sends, t->dirsend[k], t->destination[k], t->isrcsend[k], t->jsrcsend[k], t->nxsend[k], t->nysend[k]))
            return rc;
    for (int k = 0; k < t->nrecv; k++)
        if (int rc = add(L.recvs, t->dirrecv[k], t->source[k], t->idesrecv[k], t->jdesrecv[k], t->nxrecv[k], t->nyrecv[k]))
            return rc;
    // pack-buffer slots of the strided strips are handed out in TABLE order (before the sort), as the reference
    // numbers its buffers (parallel_comms_mod.f90:1664-1691); the lists are then sorted by (peer, direction)
    auto slots = [&](std::vector<Msg> &list, std::vector<Strip> &pk, long &buflen) {
        for (Msg &m : list)
            if (m.ny > 1 && m.nx != ld) {        // rows of the patch are not adjacent in memory
                m.off = buflen;
                buflen += m.count;
                if (m.count > L.max_strip) L.max_strip = (int)m.count;
            }
        std::stable_sort(list.begin(), list.end(), by_peer_dir);
        for (const Msg &m : list)
            if (m.off >= 0) pk.push_back(Strip{m.i0, m.j0, m.nx, m.ny, m.off, m.dir});
    };
    slots(L.sends, L.spack, L.sendbuf_len);
    slots(L.recvs, L.rpack, L.recvbuf_len);
    auto aggregate = [&](const std::vector<Msg> &list, std::vector<long> &agg, long &len, std::vector<Strip> &all) {
        for (const Msg &m : list) {
            agg.push_back(len);
            all.push_back(Strip{m.i0, m.j0, m.nx, m.ny, len, m.dir});
            len += (m.count + 15) & ~15L;                // every message starts on a 128-byte line
            if (m.count > L.max_msg) L.max_msg = (int)m.count;
        }
    };
    aggregate(L.sends, L.sagg, L.sagg_len, L.sall);
    aggregate(L.recvs, L.ragg, L.ragg_len, L.rall);
    return DLESM_OK;
}

// One ncclRecv / ncclSend of an exchange.  `off`: doubles from the start of the staging buffer (the receive buffer
// for a receive, the send buffer for a send), or -1 when the message travels in place, at (i0, j0) of field `field`.
struct Issue {
    bool recv;
    int peer, dir, field;          // field = -1: the message carries the strips of all nf fields, field after field
    int i0, j0, nx, ny;            // 0-based strip
    long count, off;
};

// The calls of ONE exchange of nf fields under `mask`, in issue order: the receives, then the sends (aggregated form);
// per field the receives, then the sends (single-field form: field-major).  Within each group the order of the sorted
// lists: ascending peer, then ascending direction code.  Disabled directions are left out.
static void issue_list(const std::vector<Msg> &sends, const std::vector<Msg> &recvs, const std::vector<long> &sagg,
                       const std::vector<long> &ragg, long sendbuf_len, long recvbuf_len, int nf, unsigned mask,
                       bool aggregated, std::vector<Issue> &out)
{
    out.clear();
    if (aggregated) {
        for (size_t k = 0; k < recvs.size(); k++) {
            const Msg &m = recvs[k];
            if (dir_enabled(mask, m.dir)) out.push_back(Issue{true, m.peer, m.dir, -1, m.i0, m.j0, m.nx, m.ny, nf * m.count, nf * ragg[k]});
        }
        for (size_t k = 0; k < sends.size(); k++) {
            const Msg &m = sends[k];
            if (dir_enabled(mask, m.dir)) out.push_back(Issue{false, m.peer, m.dir, -1, m.i0, m.j0, m.nx, m.ny, nf * m.count, nf * sagg[k]});
        }
        return;
    }
    for (int f = 0; f < nf; f++) {
        for (const Msg &m : recvs)
            if (dir_enabled(mask, m.dir))
                out.push_back(Issue{true, m.peer, m.dir, f, m.i0, m.j0, m.nx, m.ny, m.count, m.off >= 0 ? f * recvbuf_len + m.off : -1});
        for (const Msg &m : sends)
            if (dir_enabled(mask, m.dir))
                out.push_back(Issue{false, m.peer, m.dir, f, m.i0, m.j0, m.nx, m.ny, m.count, m.off >= 0 ? f * sendbuf_len + m.off : -1});
    }
}

// gather the strided strips into their slots: grid.y = strip, j outer / i inner
// exactly like the pack loop of parallel_comms_mod.f90:1678-1683; strips of a direction the
// mask disables are skipped (their workgroups leave at once)
__global__ void pack_strips(const double *__restrict__ f, int ld, const Strip *__restrict__ tab,
                            double *__restrict__ buf, unsigned mask)
{
    const Strip s = tab[blockIdx.y];
    if (!dir_enabled(mask, s.dir)) return;
    const long n = (long)s.nx * s.ny;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int j = (int)(t / s.nx), i = (int)(t % s.nx);
        buf[s.off + t] = f[(size_t)(s.j0 + j) * ld + s.i0 + i];
    }
}

// inverse, parallel_comms_mod.f90:1788-1793; a masked-out direction leaves its halo untouched
__global__ void unpack_strips(double *__restrict__ f, int ld, const Strip *__restrict__ tab,
                              const double *__restrict__ buf, unsigned mask)
{
    const Strip s = tab[blockIdx.y];
    if (!dir_enabled(mask, s.dir)) return;
    const long n = (long)s.nx * s.ny;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int j = (int)(t / s.nx), i = (int)(t % s.nx);
        f[(size_t)(s.j0 + j) * ld + s.i0 + i] = buf[s.off + t];
    }
}

struct FieldSet { double *f[16]; };

// aggregated forms: grid.y = message, grid.z = field; slot of field k of message s = nf*s.off + k*nx*ny
__global__ void pack_agg(FieldSet fs, int nf, int ld, const Strip *__restrict__ tab, double *__restrict__ buf, unsigned mask)
{
    const Strip s = tab[blockIdx.y];
    if (!dir_enabled(mask, s.dir)) return;
    const long n = (long)s.nx * s.ny;
    const double *__restrict__ f = fs.f[blockIdx.z];
    double *__restrict__ dst = buf + (long)nf * s.off + (long)blockIdx.z * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int j = s.ny == 1 ? 0 : (int)(t / s.nx), i = (int)(t - (long)j * s.nx);
        dst[t] = f[(size_t)(s.j0 + j) * ld + s.i0 + i];
    }
}

__global__ void unpack_agg(FieldSet fs, int nf, int ld, const Strip *__restrict__ tab, const double *__restrict__ buf,
                           unsigned mask)
{
    const Strip s = tab[blockIdx.y];
    if (!dir_enabled(mask, s.dir)) return;
    const long n = (long)s.nx * s.ny;
    double *__restrict__ f = fs.f[blockIdx.z];
    const double *__restrict__ src = buf + (long)nf * s.off + (long)blockIdx.z * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int j = s.ny == 1 ? 0 : (int)(t / s.nx), i = (int)(t - (long)j * s.nx);
        f[(size_t)(s.j0 + j) * ld + s.i0 + i] = src[t];
    }
}

extern "C" int dlesm_comm_unique_id(void *id)
{
    DLESM_REQUIRE(id != nullptr, "null id buffer");
    ncclUniqueId uid;
    DLESM_NCCL_TRY(ncclGetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
    return DLESM_OK;
}

extern "C" int dlesm_comm_init(const void *id, int nranks, int rank0)
{
    DLESM_REQUIRE(id != nullptr, "null id buffer");
    DLESM_REQUIRE(nranks >= 1 && rank0 >= 0 && rank0 < nranks, "rank %d of %d", rank0, nranks);
    DLESM_REQUIRE(g_comm == nullptr, "communicator already initialised");
    if (int rc = ensure_device()) return rc;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    DLESM_NCCL_TRY(ncclCommInitRank(&g_comm, nranks, uid, rank0));
    g_rank = rank0;
    g_size = nranks;
    return DLESM_OK;
}

extern "C" int dlesm_comm_init_mailbox(const void *id, int nranks, int rank0)
{
    DLESM_REQUIRE(id != nullptr, "null id buffer");
    DLESM_REQUIRE(nranks >= 1 && rank0 >= 0 && rank0 < nranks, "rank %d of %d", rank0, nranks);
    DLESM_REQUIRE(g_comm == nullptr && !g_mailbox, "communicator already initialised");
    if (int rc = ensure_device()) return rc;
    if (int rc = dlesm_board_open(id, nranks, rank0)) return rc;
    g_rank = rank0;
    g_size = nranks;
    g_mailbox = true;
    return DLESM_OK;
}

extern "C" int dlesm_comm_is_mailbox(void) { return g_mailbox ? 1 : 0; }

static void gather_state_release();

extern "C" int dlesm_comm_finalize(void)
{
    gather_state_release();
    if (g_mailbox) {
        (void)hipDeviceSynchronize();
        g_mailbox = false;
        g_rank = -1;
        g_size = 0;
        return dlesm_board_close();
    }
    if (g_comm) {
        (void)hipDeviceSynchronize();
        DLESM_NCCL_TRY(ncclCommDestroy(g_comm));
        g_comm = nullptr;
    }
    g_rank = -1;
    g_size = 0;
    return DLESM_OK;
}

extern "C" int dlesm_comm_rank(void) { return g_rank; }
extern "C" int dlesm_comm_size(void) { return g_size; }

static int ensure_buffers(dlesm_halo_plan *p, int nfields, hipStream_t s = nullptr);
static int peer_join(dlesm_halo_plan *p, hipStream_t s);

extern "C" int dlesm_halo_plan_create(const dlesm_comm_tables *t, int ld, int ny, dlesm_halo_plan **out)
{
    DLESM_REQUIRE(t != nullptr && out != nullptr, "null pointer");
    MsgLists L;
    if (int rc0 = build_msg_lists(t, ld, ny, L)) return rc0;
    if (int rc0 = ensure_device()) return rc0;
    dlesm_halo_plan *p = new dlesm_halo_plan;
    p->ld = ld;
    p->ny = ny;
    p->sends = L.sends;
    p->recvs = L.recvs;
    p->sendbuf_len = L.sendbuf_len;
    p->recvbuf_len = L.recvbuf_len;
    p->max_strip = L.max_strip;
    p->n_spack = (int)L.spack.size();
    p->n_rpack = (int)L.rpack.size();
    p->sagg = L.sagg;
    p->ragg = L.ragg;
    p->sagg_len = L.sagg_len;
    p->ragg_len = L.ragg_len;
    p->max_msg = L.max_msg;
    const std::vector<Strip> &spack = L.spack, &rpack = L.rpack, &sall = L.sall, &rall = L.rall;
    int rc = DLESM_OK;
    auto upload = [&](const std::vector<Strip> &v, Strip **d) -> int {
        if (v.empty()) return DLESM_OK;
        DLESM_HIP_TRY(hipMalloc((void **)d, v.size() * sizeof(Strip)));
        DLESM_HIP_TRY(hipMemcpy(*d, v.data(), v.size() * sizeof(Strip), hipMemcpyHostToDevice));
        return DLESM_OK;
    };
    if ((rc = upload(spack, &p->d_spack)) || (rc = upload(rpack, &p->d_rpack)) || (rc = upload(sall, &p->d_sall)) ||
        (rc = upload(rall, &p->d_rall))) {
        dlesm_halo_plan_destroy(p);
        return rc;
    }
    // persistent buffers: allocated once, unlike recvBuff which the reference
    // reallocates on every exchange (pcomms:1574-1592,1852)
    if ((rc = ensure_buffers(p, 1))) {
        dlesm_halo_plan_destroy(p);
        return rc;
    }
    // The events order work of two streams of ONE device: no system-scope fence (L2 write-back) is
    // needed when they are recorded -- 3 us less per step on the caller's stream (scripts/syncbench.hip)
    const unsigned evflags = hipEventDisableTiming | (tuning("dm_event_system_fence", 0) ? 0u : hipEventDisableSystemFence);
    if (hipEventCreateWithFlags(&p->ev_frame, evflags) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_comm, evflags) != hipSuccess) {
        dlesm_halo_plan_destroy(p);
        return fail(DLESM_EHIP, "halo plan: hipEventCreate failed");
    }
    if (!p->sends.empty() || !p->recvs.empty()) {
        void *words = nullptr;
        if (hipMalloc(&words, 256) != hipSuccess || hipMemset(words, 0, 256) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            dlesm_halo_plan_destroy(p);
            return fail(DLESM_EHIP, "halo plan: cannot allocate the frame flag");
        }
        p->frame_flag = (unsigned long long *)words;                 // three words, 64 bytes apart
        p->frame_counter = (unsigned *)((char *)words + 64);
        p->halo_flag = (unsigned long long *)((char *)words + 128);
        p->frame_timed_out = wait_timed_out_word();         // ONE word for the process: sticky, checked by every entry
        // the probe the one-launch forms rest on (kernels of two streams side by side), once, here, for the stream most
        // programs use -- not inside somebody's first time step
        (void)streams_run_concurrently(nullptr);
    }
    if (g_mailbox && g_size > 1) {      // COLLECTIVE in mailbox mode: every rank creates its plans at the same points
        int fcap = tuning("mailbox_fields", 3);
        if (const char *e = getenv("DLESM_MAILBOX_FIELDS")) fcap = atoi(e);
        if (fcap < 1 || fcap > 16) fcap = 3;
        std::vector<char> all((size_t)g_size * DLESM_PEER_BLOB_BYTES);
        char mine[DLESM_PEER_BLOB_BYTES];
        if ((rc = dlesm_halo_plan_peer_export(p, g_rank, fcap, mine)) || (rc = dlesm_board_allgather(mine, sizeof mine, all.data())) ||
            (rc = dlesm_halo_plan_peer_connect(p, g_rank, g_size, all.data()))) {
            dlesm_halo_plan_destroy(p);
            return rc;
        }
    }
    *out = p;
    return DLESM_OK;
}

extern "C" int dlesm_halo_plan_destroy(dlesm_halo_plan *p)
{
    if (!p) return DLESM_OK;
    // a peer-transport step still pending: its join is also what tells that the neighbours are done storing into this
    // rank's mailbox, which is about to be freed
    if (p->peer_pending && p->pending_field) (void)peer_join(p, p->pending_stream);
    (void)hipDeviceSynchronize();
    if (p->d_spack) (void)hipFree(p->d_spack);
    if (p->d_rpack) (void)hipFree(p->d_rpack);
    if (p->d_sall) (void)hipFree(p->d_sall);
    if (p->d_rall) (void)hipFree(p->d_rall);
    if (p->sendagg) (void)hipFree(p->sendagg);
    if (p->recvagg) (void)hipFree(p->recvagg);
    if (p->sendbuf) (void)hipFree(p->sendbuf);
    if (p->recvbuf) (void)hipFree(p->recvbuf);
    if (p->frame_flag) (void)hipFree(p->frame_flag);
    for (void *m : p->peer_mapped) (void)hipIpcCloseMemHandle(m);
    if (p->peer_box) (void)hipFree(p->peer_box);
    if (p->peer_counter) (void)hipFree(p->peer_counter);
    if (p->ev_peer) (void)hipEventDestroy(p->ev_peer);
    if (p->ev_frame) (void)hipEventDestroy(p->ev_frame);
    if (p->ev_comm) (void)hipEventDestroy(p->ev_comm);
    delete p;
    return DLESM_OK;
}

// What a plan made from `tables` for fields of ld x ny hands to RCCL in ONE exchange, call by call, in issue order.
// Host only: no device, no communicator -- the lists are built by the code dlesm_halo_plan_create uses and walked by
// the code the exchanges use.
extern "C" int dlesm_halo_plan_describe(const dlesm_comm_tables *tables, int ld, int ny, int nfields, unsigned dirs_mask,
                                        int aggregated, dlesm_msg_desc *out, int max_out, int *n_out)
{
    DLESM_REQUIRE(tables != nullptr && n_out != nullptr && (out != nullptr || max_out == 0), "null pointer");
    DLESM_REQUIRE(nfields >= 1 && nfields <= 16, "%d fields", nfields);
    MsgLists L;
    if (int rc = build_msg_lists(tables, ld, ny, L)) return rc;
    std::vector<Issue> calls;
    issue_list(L.sends, L.recvs, L.sagg, L.ragg, L.sendbuf_len, L.recvbuf_len, nfields, dirs_mask, aggregated != 0, calls);
    *n_out = (int)calls.size();
    DLESM_REQUIRE((int)calls.size() <= max_out || max_out == 0, "%zu calls, room for %d", calls.size(), max_out);
    for (size_t k = 0; k < calls.size() && (int)k < max_out; k++) {
        const Issue &c = calls[k];
        out[k] = dlesm_msg_desc{c.recv ? 1 : 0, c.peer, c.dir, c.field, c.i0 + 1, c.j0 + 1, c.nx, c.ny, c.count, c.off};
    }
    return DLESM_OK;
}

// A stream that is being captured into a hipGraph (hipStreamBeginCapture): the steps then take
// their event form only -- fork to the side stream and join back are graph edges, every replay
// runs the same nodes -- because the one-launch forms hand over through sequence numbers that are
// baked into kernel arguments and advance per call on the host.
static bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// Pack buffers hold one slot set per field of a multi-field exchange.
static int ensure_buffers(dlesm_halo_plan *p, int nfields, hipStream_t s)
{
    if (nfields <= p->buf_fields) return DLESM_OK;
    DLESM_REQUIRE(!s || !capturing(s), "pack buffers for %d field(s) are not allocated yet: run the call once before "
                  "capturing it into a graph", nfields);
    DLESM_HIP_TRY(hipDeviceSynchronize());               // nobody may still be using the old ones
    if (p->sendbuf) DLESM_HIP_TRY(hipFree(p->sendbuf));
    if (p->recvbuf) DLESM_HIP_TRY(hipFree(p->recvbuf));
    p->sendbuf = p->recvbuf = nullptr;
    if (p->sendbuf_len) DLESM_HIP_TRY(hipMalloc((void **)&p->sendbuf, (size_t)nfields * p->sendbuf_len * sizeof(double)));
    if (p->recvbuf_len) DLESM_HIP_TRY(hipMalloc((void **)&p->recvbuf, (size_t)nfields * p->recvbuf_len * sizeof(double)));
    p->buf_fields = nfields;
    return DLESM_OK;
}

// Checked by every entry that may put RCCL calls into a capture, BEFORE it enqueues anything (a refusal
// then leaves the caller's capture intact).  Measured (scripts/graphprobe.hip): a captured
// ncclSend/ncclRecv group is fine with RCCL 2.27.7 / HIP 7.2 and segfaults inside hipStreamEndCapture
// with the RCCL 2.26.6 / HIP 7.0 pair that PyTorch 2.10 bundles.
// over_mailboxes: the entry will take the peer transport, which has no RCCL call to capture (its sequence numbers live on the
// device, peer_seq_load) -- only an EVEN number of mailbox operations per plan and graph is asked for.
static int capture_ok(const dlesm_halo_plan *p, hipStream_t s, bool over_mailboxes = false)
{
    if (over_mailboxes || (p->sends.empty() && p->recvs.empty()) || !capturing(s)) return DLESM_OK;
    int v = 0;
    DLESM_NCCL_TRY(ncclGetVersion(&v));
    DLESM_REQUIRE(v >= 22707 || tuning("dm_graph_force", 0),
                  "RCCL %d.%d.%d in this process cannot be captured into a hipGraph (needs >= 2.27.7): issue the "
                  "exchange outside the capture", v / 10000, v / 100 % 100, v % 100);
    return DLESM_OK;
}

// A pipelined step leaves its exchange in flight: whoever touches the plan or the halos next on
// stream `s` is ordered behind it first.
static int join_pending(dlesm_halo_plan *p, hipStream_t s)
{
    if (p->peer_pending)
        if (int rc = peer_join(p, s)) return rc;
    if (!p->pending) return DLESM_OK;
    DLESM_REQUIRE(!capturing(s), "a pipelined step is in flight: call dlesm_halo_plan_join before capturing a graph");
    DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
    p->pending = false;
    if (p->pending_field && p->n_rpack) {      // the unpack the pipelined step left out
        int gx = (p->max_strip + 255) / 256;
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(unpack_strips, dim3(gx, p->n_rpack), dim3(256), 0, s, p->pending_field, p->ld, p->d_rpack,
                           p->recvbuf, p->pending_mask);
        DLESM_HIP_TRY(hipGetLastError());
    }
    p->pending_field = nullptr;
    return DLESM_OK;
}

extern "C" int dlesm_halo_plan_join(dlesm_halo_plan *p, void *stream)
{
    DLESM_REQUIRE(p != nullptr, "null plan");
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "a distributed step gave up waiting for a flag (frame or halo wait timed out)");
    return join_pending(p, (hipStream_t)stream);
}

static int ensure_agg(dlesm_halo_plan *p, int nfields, hipStream_t s)
{
    if (nfields <= p->agg_fields) return DLESM_OK;
    DLESM_REQUIRE(!s || !capturing(s), "aggregated buffers for %d fields are not allocated yet: run the call once "
                  "before capturing it into a graph", nfields);
    DLESM_HIP_TRY(hipDeviceSynchronize());               // nobody may still be using the old ones
    if (p->sendagg) DLESM_HIP_TRY(hipFree(p->sendagg));
    if (p->recvagg) DLESM_HIP_TRY(hipFree(p->recvagg));
    p->sendagg = p->recvagg = nullptr;
    if (p->sagg_len) DLESM_HIP_TRY(hipMalloc((void **)&p->sendagg, (size_t)nfields * p->sagg_len * sizeof(double)));
    if (p->ragg_len) DLESM_HIP_TRY(hipMalloc((void **)&p->recvagg, (size_t)nfields * p->ragg_len * sizeof(double)));
    p->agg_fields = nfields;
    return DLESM_OK;
}

// nf > 1 fields, ONE message per neighbour and direction (see dlesm_halo_plan::sagg).  Between a pair of
// ranks messages match in issue order: ascending direction code on both sides.  `prepacked`: the caller's
// kernel has already written every enabled strip of every field into the aggregated send buffer.
static int exchange_agg(dlesm_halo_plan *p, double *const *fields, int nf, unsigned mask, hipStream_t s, bool prepacked)
{
    bool any = false;
    for (const Msg &m : p->sends) any |= dir_enabled(mask, m.dir);
    for (const Msg &m : p->recvs) any |= dir_enabled(mask, m.dir);
    if (!any) return DLESM_OK;
    DLESM_REQUIRE(g_comm != nullptr, "halo exchange before dlesm_comm_init");
    DLESM_REQUIRE(nf <= 16, "at most 16 fields per aggregated exchange");
    if (int rc = capture_ok(p, s)) return rc;
    if (int rc = ensure_agg(p, nf, s)) return rc;
    FieldSet fs{};
    for (int k = 0; k < nf; k++) fs.f[k] = fields[k];
    int gx = (p->max_msg + 255) / 256;
    if (gx > 32) gx = 32;
    if (!prepacked && !p->sends.empty())
        hipLaunchKernelGGL(pack_agg, dim3(gx, (unsigned)p->sends.size(), nf), dim3(256), 0, s, fs, nf, p->ld, p->d_sall,
                           p->sendagg, mask);
    const int skip = tuning("dm_skip_parts", 0);         // diagnostics, see exchange_on
    if (!(skip & 1)) {
        DLESM_NCCL_TRY(ncclGroupStart());
        ncclResult_t err = ncclSuccess;
        std::vector<Issue> calls;
        issue_list(p->sends, p->recvs, p->sagg, p->ragg, p->sendbuf_len, p->recvbuf_len, nf, mask, true, calls);
        for (const Issue &c : calls) {
            if (c.recv) DLESM_NCCL_IN_GROUP(err, ncclRecv(p->recvagg + c.off, (size_t)c.count, ncclDouble, c.peer, g_comm, s));
            else DLESM_NCCL_IN_GROUP(err, ncclSend(p->sendagg + c.off, (size_t)c.count, ncclDouble, c.peer, g_comm, s));
        }
        if (int rc = group_end(err, "aggregated halo exchange (ncclSend/ncclRecv)")) return rc;
    }
    if (!(skip & 2) && !p->recvs.empty())
        hipLaunchKernelGGL(unpack_agg, dim3(gx, (unsigned)p->recvs.size(), nf), dim3(256), 0, s, fs, nf, p->ld, p->d_rall,
                           p->recvagg, mask);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

// The exchange of one field in the Jacobi step's own form (rows sent and received in place, only the
// strided strips packed; nf > 1 kept as the comparison point of exchange_agg, dm_aggregate = 0): between a
// pair of ranks messages match in issue order, field-major, then ascending direction code, on both sides.
// `prepacked`: the caller's kernel has already written the enabled strided strips of every field
// into the send buffer (dlesm_jacobi5_step_dm's frame kernel does), so no pack launch is needed.
static int peer_in_strips(const dlesm_halo_plan *p, unsigned mask, unsigned long long seq, int nf, PeerJob::In *in, int *n);
static int peer_order(dlesm_halo_plan *p, hipStream_t s);

// r2d_field%halo_exchange over the mailboxes (a connected plan with room for nf fields): TWO small launches on the caller's
// stream -- every enabled send strip copied into the neighbours' mailboxes and their arrival flags raised; then the wait
// for this rank's own flags and the copy of the received strips into the halos -- instead of pack + RCCL group + unpack.
// What MPI_Isend / MPI_Irecv / MPI_Waitany + the unpack loop do in the reference (parallel_comms_mod.f90:1601-1798).
// Send strips are read from the internal region only, so no message depends on another one's arrival (any depth).
static int exchange_peer(dlesm_halo_plan *p, double *const *fields, int nf, unsigned mask, hipStream_t s)
{
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier wait for a neighbour's arrival flag gave up");
    if (int rc = peer_order(p, s)) return rc;
    const unsigned long long seq = p->peer_seq + 1;
    PeerOuts out{};
    for (size_t m = 0; m < p->sends.size(); m++) {
        const Msg &sm = p->sends[m];
        DLESM_REQUIRE(out.n < PeerJob::MAXM, "more than %d send messages in one exchange", PeerJob::MAXM);
        // a direction the mask disables sends NO cells but still raises its flag (an empty strip): see peer_in_strips
        const bool on = dir_enabled(mask, sm.dir);
        out.s[out.n++] = PeerJob::Out{sm.i0, sm.j0, on ? sm.nx : 0, on ? sm.ny : 0,
                                      p->peer_tx[m] + (seq & 1) * p->peer_tx_par[m] + (long)nf * p->peer_tx_off[m], p->peer_txflag[m]};
    }
    PeerStrips st{};
    if (int rc = peer_in_strips(p, mask, seq, nf, st.s, &st.n)) return rc;
    p->peer_seq = seq;
    if (tuning("dm_peer_one_launch", 1))       // both halves in one launch (0: a pack launch, then a wait + unpack launch)
        return launch_peer_exchange(out, st, fields, nf, p->ld, p->peer_counter, seq, p->peer_seqw, p->frame_timed_out, s);
    if (int rc = launch_peer_pack(out, fields, nf, p->ld, p->peer_counter, seq, p->peer_seqw, p->frame_timed_out, s)) return rc;
    return launch_peer_unpack(st, seq, p->peer_seqw, fields, nf, p->ld, p->frame_timed_out, s);
}

static int exchange_on(dlesm_halo_plan *p, double *const *fields, int nf, unsigned mask, hipStream_t s,
                       bool prepacked = false, bool skip_unpack = false)
{
    bool any = false, any_spack = false, any_rpack = false;
    for (const Msg &m : p->sends)
        if (dir_enabled(mask, m.dir)) { any = true; any_spack |= m.off >= 0; }
    for (const Msg &m : p->recvs)
        if (dir_enabled(mask, m.dir)) { any = true; any_rpack |= m.off >= 0; }
    if (!any) return DLESM_OK; // serial run, or no direction enabled: nothing to do (pcomms:1546,1557-1571)
    if (p->peer_on && nf <= p->peer_fcap && !prepacked && !skip_unpack &&
        (g_mailbox || (tuning("dm_peer", 1) && tuning("dm_peer_exchange", 1))))      // (mailbox mode has no other transport)
        return exchange_peer(p, fields, nf, mask, s);
    // (a single field on its own goes the same way: rows staged through the buffer travel faster than rows
    //  sent in place from their 8-byte-aligned position in the field -- 42 against 52 us at 8192^2)
    if ((nf > 1 || (!prepacked && !skip_unpack && tuning("dm_aggregate_single", 1))) && tuning("dm_aggregate", 1))
        return exchange_agg(p, fields, nf, mask, s, prepacked);
    DLESM_REQUIRE(g_comm != nullptr, "halo exchange before dlesm_comm_init");
    if (int rc = capture_ok(p, s)) return rc;
    if (int rc = ensure_buffers(p, nf, s)) return rc;
    int gx = (p->max_strip + 255) / 256;
    if (gx > 64) gx = 64;
    if (any_spack && !prepacked)
        for (int k = 0; k < nf; k++)
            hipLaunchKernelGGL(pack_strips, dim3(gx, p->n_spack), dim3(256), 0, s, fields[k], p->ld, p->d_spack,
                               p->sendbuf + (size_t)k * p->sendbuf_len, mask);
    // diagnostics (profiling only, results are then wrong): price the parts of an exchange
    const int skip = tuning("dm_skip_parts", 0);         // bit0: no RCCL group, bit1: no unpack
    if (skip & 2) skip_unpack = true;
    if (!(skip & 1)) {
        DLESM_NCCL_TRY(ncclGroupStart());
        ncclResult_t err = ncclSuccess;
        std::vector<Issue> calls;
        issue_list(p->sends, p->recvs, p->sagg, p->ragg, p->sendbuf_len, p->recvbuf_len, nf, mask, false, calls);
        for (const Issue &c : calls) {
            double *inplace = fields[c.field] + (size_t)c.j0 * p->ld + c.i0;
            if (c.recv) DLESM_NCCL_IN_GROUP(err, ncclRecv(c.off >= 0 ? p->recvbuf + c.off : inplace, (size_t)c.count, ncclDouble, c.peer, g_comm, s));
            else DLESM_NCCL_IN_GROUP(err, ncclSend(c.off >= 0 ? p->sendbuf + c.off : inplace, (size_t)c.count, ncclDouble, c.peer, g_comm, s));
        }
        if (int rc = group_end(err, "halo exchange (ncclSend/ncclRecv)")) return rc;
    }
    if (any_rpack && !skip_unpack)
        for (int k = 0; k < nf; k++)
            hipLaunchKernelGGL(unpack_strips, dim3(gx, p->n_rpack), dim3(256), 0, s, fields[k], p->ld, p->d_rpack,
                               p->recvbuf + (size_t)k * p->recvbuf_len, mask);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

static int exchange_on(dlesm_halo_plan *p, double *f, unsigned mask, hipStream_t s, bool prepacked = false,
                       bool skip_unpack = false)
{
    double *one[1] = {f};
    return exchange_on(p, one, 1, mask, s, prepacked, skip_unpack);
}

extern "C" int dlesm_halo_exchange_multi_f64(dlesm_halo_plan *p, double *const *fields, int nfields,
                                             unsigned dirs_mask, void *stream)
{
    DLESM_REQUIRE(p != nullptr && fields != nullptr && nfields >= 1 && nfields <= 16, "bad arguments");
    for (int k = 0; k < nfields; k++) DLESM_REQUIRE(fields[k] != nullptr, "null field %d", k);
    if (int rc = ensure_device()) return rc;
    if (int rc = join_pending(p, (hipStream_t)stream)) return rc;
    if (g_mailbox && p->peer_on && nfields > p->peer_fcap) {      // more fields than a mailbox has room for: in turns
        for (int k = 0; k < nfields; k += p->peer_fcap) {
            const int nf = std::min(p->peer_fcap, nfields - k);
            if (int rc = exchange_on(p, fields + k, nf, dirs_mask & 0x1Fu, (hipStream_t)stream)) return rc;
        }
        return DLESM_OK;
    }
    return exchange_on(p, fields, nfields, dirs_mask & 0x1Fu, (hipStream_t)stream);
}

extern "C" int dlesm_halo_exchange_f64(dlesm_halo_plan *p, double *field, unsigned dirs_mask, void *stream)
{
    DLESM_REQUIRE(p != nullptr && field != nullptr, "null pointer");
    if (int rc = ensure_device()) return rc;
    if (int rc = join_pending(p, (hipStream_t)stream)) return rc;
    return exchange_on(p, field, dirs_mask & 0x1Fu, (hipStream_t)stream);
}

// hipIpcOpenMemHandle, ONE IMPORTER AT A TIME.
// Round 3 saw a mapping fail once in dozens of runs, only in jobs where three or more processes opened the SAME handle (the
// root's gather buffer) at the same moment, and succeed when tried again.  With HSA_ENABLE_IPC_MODE_LEGACY=0 -- the only mode
// this pool's driver supports -- ROCr hands the dmabuf of an allocation from the exporting to the importing process over a
// unix socket (its imports: socket / listen / accept / connect / sendmsg / recvmsg), i.e. every importer makes a connection
// to the exporter, and several at once are what the failure needs.  So importers of one node take turns: an exclusive
// flock(2) on a per-user file in /dev/shm around the call (any transport mode, no extra round of the board; a mapping takes
// well under a millisecond and happens at plan creation and at the first gather only, see ipc_cached_open).
// What is left of round 3's retry loop is a LAST RESORT that is never silent: every failed attempt is logged with its error
// name and counted (dlesm_ipc_open_retries), and the multi-process tests require the count to be zero.
namespace {
std::atomic<int> g_ipc_retries{0};
struct IpcLock {
    int fd = -1;
    IpcLock()
    {
        char name[96];
        snprintf(name, sizeof name, "/dev/shm/dlesm-ipc-%u.lock", (unsigned)getuid());
        fd = open(name, O_RDWR | O_CREAT | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd >= 0 && flock(fd, LOCK_EX) != 0) { close(fd); fd = -1; }
        if (fd < 0) {
            static bool said = false;
            if (!said) fprintf(stderr, "dlesm: cannot take %s (%s): IPC mappings are opened without taking turns\n", name, strerror(errno));
            said = true;
        }
    }
    ~IpcLock() { if (fd >= 0) { flock(fd, LOCK_UN); close(fd); } }
};
} // namespace

static hipError_t ipc_open(void **ptr, hipIpcMemHandle_t handle)
{
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 3; attempt++) {
        {
            IpcLock turn;
            e = hipIpcOpenMemHandle(ptr, handle, hipIpcMemLazyEnablePeerAccess);
        }
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        g_ipc_retries++;
        fprintf(stderr, "dlesm: rank %d pid %ld: hipIpcOpenMemHandle attempt %d FAILED with %s (%s)%s -- counted, see dlesm_ipc_open_retries\n",
                g_rank, (long)getpid(), attempt + 1, hipGetErrorName(e), hipGetErrorString(e), attempt < 2 ? ", trying again" : "");
        const struct timespec nap = {0, (8L << attempt) * 1000 * 1000};
        nanosleep(&nap, nullptr);
    }
    return e;
}

extern "C" int dlesm_ipc_open_retries(void) { return g_ipc_retries.load(); }

// The gather buffer of mailbox mode (gather_mailbox): OWNED BY THE LIBRARY on the root, exported once, mapped once per job by
// every other rank -- not the caller's receive buffer opened and closed around every gather as in round 3 (a storm of
// concurrent imports of one handle per output step, and a mapping whose lifetime was the caller's business).  The root
// re-allocates it only to grow and counts the generations; a rank that sees a new generation closes its old mapping first.
namespace {
void *g_gather_buf = nullptr;            // root
size_t g_gather_cap = 0;
unsigned long long g_gather_gen = 0;
hipIpcMemHandle_t g_gather_handle;
void *g_gather_map = nullptr;            // other ranks: the root's buffer, mapped
unsigned long long g_gather_map_gen = 0;
} // namespace

static void gather_state_release()
{
    if (g_gather_map) (void)hipIpcCloseMemHandle(g_gather_map);
    if (g_gather_buf) (void)hipFree(g_gather_buf);
    g_gather_map = g_gather_buf = nullptr;
    g_gather_cap = 0;
    g_gather_map_gen = 0;
}

// ---- peer transport ------------------------------------------------------------------------------------------------
// What one rank tells the others (dlesm_halo_plan_peer_export), DLESM_PEER_BLOB_BYTES per rank, all-gathered by the host
// program or by dlesm_halo_plan_peer_connect_rccl: where its mailbox is (an IPC handle) and which slot and flag each of
// its receive messages has, in the plan's (peer, direction) order -- the k-th message a rank sends to a neighbour is the
// k-th that neighbour receives from it, the matching rule of the RCCL path.
namespace {
struct PeerBlob {
    char magic[8];
    int rank, nrecv, fcap, has_handle;
    long par_len;
    long pid;
    hipIpcMemHandle_t handle;
    struct R { int peer, dir; long count, off; } r[16];
};
static_assert(sizeof(PeerBlob) <= DLESM_PEER_BLOB_BYTES, "peer blob size");
constexpr char PEER_MAGIC[8] = {'D', 'L', 'E', 'S', 'M', 'P', 'B', '1'};
constexpr size_t PEER_PAYLOAD_AT = 4096;
} // namespace

// ---- the host-only part of the peer transport: which receive of which neighbour every send meets -----------------
// dlesm_halo_plan_peer_export / _connect use exactly these two functions; dlesm_peer_match_describe runs them on bare tables
// (no device, no mailbox) so that the matching can be checked for any mesh on a machine without a GPU.
static void peer_blob_fill(PeerBlob &b, int my_rank, int fcap, long ragg_len, const std::vector<Msg> &recvs, const std::vector<long> &ragg)
{
    memset(&b, 0, sizeof b);
    memcpy(b.magic, PEER_MAGIC, 8);
    b.rank = my_rank;
    b.nrecv = (int)recvs.size();
    b.fcap = fcap;
    b.par_len = ((long)fcap * ragg_len + 15) & ~15L;
    b.pid = (long)getpid();
    for (size_t m = 0; m < recvs.size(); m++) b.r[m] = PeerBlob::R{recvs[m].peer, recvs[m].dir, recvs[m].count, ragg[m]};
}

struct PeerMatch { int peer, slot; long off, par_len; };

static int peer_match(const std::vector<Msg> &sends, int my_rank, int nranks, int fcap, const void *blobs, std::vector<PeerMatch> &out)
{
    auto blob_of = [&](int r) { PeerBlob b; memcpy(&b, (const char *)blobs + (size_t)r * DLESM_PEER_BLOB_BYTES, sizeof b); return b; };
    out.assign(sends.size(), PeerMatch{-1, -1, 0, 0});
    for (size_t m = 0; m < sends.size(); m++) {
        const Msg &sm = sends[m];
        DLESM_REQUIRE(sm.peer >= 0 && sm.peer < nranks, "send %zu goes to rank %d of %d", m, sm.peer, nranks);
        const PeerBlob b = blob_of(sm.peer);
        DLESM_REQUIRE(memcmp(b.magic, PEER_MAGIC, 8) == 0 && b.rank == sm.peer && b.nrecv >= 0 && b.nrecv <= 16,
                      "peer blob of rank %d is not valid", sm.peer);
        DLESM_REQUIRE(b.fcap == fcap, "rank %d made its mailbox for %d field(s), this rank for %d", sm.peer, b.fcap, fcap);
        int k = 0;                                      // this is my k-th message to that neighbour ...
        for (size_t q = 0; q < m; q++) k += sends[q].peer == sm.peer;
        int j = -1;                                     // ... and meets its k-th receive from me
        for (int q = 0, seen = 0; q < b.nrecv; q++)
            if (b.r[q].peer == my_rank && seen++ == k) { j = q; break; }
        DLESM_REQUIRE(j >= 0 && b.r[j].count == sm.count, "send %zu (to rank %d, direction %d, %ld cells) has no matching receive there",
                      m, sm.peer, sm.dir, sm.count);
        out[m] = PeerMatch{sm.peer, j, b.r[j].off, b.par_len};
    }
    return DLESM_OK;
}

// Host only (no device, no communicator): the blob a plan made from `tables` for ld x ny fields would export (without an IPC
// handle), and, given the blobs of all ranks, the (neighbour, receive index there) every send of this rank is matched with.
extern "C" int dlesm_peer_blob_describe(const dlesm_comm_tables *tables, int ld, int ny, int my_rank, int nfields, void *blob)
{
    DLESM_REQUIRE(tables != nullptr && blob != nullptr && nfields >= 1 && nfields <= 16, "bad arguments");
    MsgLists L;
    if (int rc = build_msg_lists(tables, ld, ny, L)) return rc;
    DLESM_REQUIRE(L.recvs.size() <= 16 && L.sends.size() <= 16, "more than 16 messages");
    PeerBlob b;
    peer_blob_fill(b, my_rank, nfields, L.ragg_len, L.recvs, L.ragg);
    memset(blob, 0, DLESM_PEER_BLOB_BYTES);
    memcpy(blob, &b, sizeof b);
    return DLESM_OK;
}

extern "C" int dlesm_peer_match_describe(const dlesm_comm_tables *tables, int ld, int ny, int my_rank, int nranks, int nfields,
                                         const void *blobs, dlesm_peer_match_desc *out, int max_out, int *n_out)
{
    DLESM_REQUIRE(tables != nullptr && blobs != nullptr && n_out != nullptr && (out != nullptr || max_out == 0), "null pointer");
    MsgLists L;
    if (int rc = build_msg_lists(tables, ld, ny, L)) return rc;
    std::vector<PeerMatch> mt;
    if (int rc = peer_match(L.sends, my_rank, nranks, nfields, blobs, mt)) return rc;
    *n_out = (int)mt.size();
    DLESM_REQUIRE((int)mt.size() <= max_out || max_out == 0, "%zu sends, room for %d", mt.size(), max_out);
    for (size_t m = 0; m < mt.size() && (int)m < max_out; m++) {
        const Msg &sm = L.sends[m];
        out[m] = dlesm_peer_match_desc{sm.peer, sm.dir, sm.i0 + 1, sm.j0 + 1, sm.nx, sm.ny, sm.count, mt[m].slot, mt[m].off};
    }
    return DLESM_OK;
}

extern "C" int dlesm_halo_plan_peer_export(dlesm_halo_plan *p, int my_rank, int nfields, void *blob)
{
    DLESM_REQUIRE(p != nullptr && blob != nullptr, "null pointer");
    DLESM_REQUIRE(nfields >= 1 && nfields <= 16, "%d fields", nfields);
    DLESM_REQUIRE(p->recvs.size() <= 16 && p->sends.size() <= 16, "more than 16 messages");
    if (int rc = ensure_device()) return rc;
    DLESM_REQUIRE(!p->peer_on, "the plan is already connected to its peers");
    if (!p->peer_box) {
        p->peer_fcap = nfields;
        p->peer_par_len = ((long)nfields * p->ragg_len + 15) & ~15L;
        const size_t bytes = PEER_PAYLOAD_AT + 2 * (size_t)p->peer_par_len * sizeof(double) + 256;
        // UNCACHED device memory (MTYPE UC, what RCCL takes for its own flags and FIFOs): no line of it ever sits in an L2
        // on either side, so stores arriving over xGMI and this GPU's own loads meet in memory while kernels run --
        // ordinary (coarse-grained) memory is only coherent across GPUs at kernel boundaries.  mailbox_finegrained = 1
        // asks for the fine-grained pool instead.
        if (tuning("mailbox_finegrained", 0) ||
            hipExtMallocWithFlags(&p->peer_box, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            p->peer_box = nullptr;
            DLESM_HIP_TRY(hipExtMallocWithFlags(&p->peer_box, bytes, hipDeviceMallocFinegrained));
        }
        DLESM_HIP_TRY(hipMemset(p->peer_box, 0, bytes));
        DLESM_HIP_TRY(hipMalloc((void **)&p->peer_counter, 128));
        DLESM_HIP_TRY(hipMemset(p->peer_counter, 0, 128));
        p->peer_seqw = (unsigned long long *)(p->peer_counter + 16);      // its own 64 bytes
        const unsigned long long first[3] = {0, 1, 0};                   // next even / next odd operation (the first is number 1) / last raised
        DLESM_HIP_TRY(hipMemcpy(p->peer_seqw, first, sizeof first, hipMemcpyHostToDevice));
        DLESM_HIP_TRY(hipDeviceSynchronize());
        p->peer_flags = (unsigned long long *)p->peer_box;
        p->peer_rx = (double *)((char *)p->peer_box + PEER_PAYLOAD_AT);
        DLESM_HIP_TRY(hipEventCreateWithFlags(&p->ev_peer, hipEventDisableTiming));
    }
    DLESM_REQUIRE(nfields == p->peer_fcap, "mailbox was made for %d field(s)", p->peer_fcap);
    PeerBlob b;
    peer_blob_fill(b, my_rank, p->peer_fcap, p->ragg_len, p->recvs, p->ragg);
    // (a process cannot open its own handle: a rank that is its own neighbour -- loop-back -- uses the pointer)
    b.has_handle = hipIpcGetMemHandle(&b.handle, p->peer_box) == hipSuccess ? 1 : 0;
    if (!b.has_handle) (void)hipGetLastError();
    memset(blob, 0, DLESM_PEER_BLOB_BYTES);
    memcpy(blob, &b, sizeof b);
    return DLESM_OK;
}

extern "C" int dlesm_halo_plan_peer_connect(dlesm_halo_plan *p, int my_rank, int nranks, const void *blobs)
{
    DLESM_REQUIRE(p != nullptr && blobs != nullptr, "null pointer");
    DLESM_REQUIRE(p->peer_box != nullptr, "call dlesm_halo_plan_peer_export first");
    DLESM_REQUIRE(!p->peer_on, "the plan is already connected to its peers");
    DLESM_REQUIRE(my_rank >= 0 && my_rank < nranks, "rank %d of %d", my_rank, nranks);
    if (int rc = ensure_device()) return rc;
    auto blob_of = [&](int r) { PeerBlob b; memcpy(&b, (const char *)blobs + (size_t)r * DLESM_PEER_BLOB_BYTES, sizeof b); return b; };
    std::vector<PeerMatch> mt;                            // which receive of which neighbour every send meets: host-only code,
    if (int rc0 = peer_match(p->sends, my_rank, nranks, p->peer_fcap, blobs, mt)) return rc0;   // also behind dlesm_peer_match_describe
    std::vector<void *> base(nranks, nullptr);
    const size_t ns = p->sends.size();
    std::vector<double *> tx(ns);
    std::vector<long> par(ns), off(ns);
    std::vector<unsigned long long *> txf(ns);
    std::vector<void *> mapped;
    int rc = DLESM_OK;
    for (size_t m = 0; m < ns && !rc; m++) {
        const int peer = mt[m].peer;
        if (!base[peer]) {
            if (peer == my_rank) {
                base[peer] = p->peer_box;
            } else {
                const PeerBlob b = blob_of(peer);
                if (!b.has_handle) { rc = fail(DLESM_EHIP, "rank %d could not export its mailbox (hipIpcGetMemHandle failed there)", peer); break; }
                void *ptr = nullptr;
                const hipError_t e = ipc_open(&ptr, b.handle);
                if (e != hipSuccess) { rc = fail(DLESM_EHIP, "hipIpcOpenMemHandle(mailbox of rank %d): %s", peer, hipGetErrorString(e)); break; }
                mapped.push_back(ptr);
                base[peer] = ptr;
            }
        }
        tx[m] = (double *)((char *)base[peer] + PEER_PAYLOAD_AT);
        par[m] = mt[m].par_len;
        off[m] = mt[m].off;
        txf[m] = (unsigned long long *)base[peer] + mt[m].slot;
    }
    if (rc) {
        for (void *q : mapped) (void)hipIpcCloseMemHandle(q);
        return rc;
    }
    p->peer_tx = tx, p->peer_tx_par = par, p->peer_tx_off = off, p->peer_txflag = txf, p->peer_mapped = mapped;
    p->peer_on = true;
    return DLESM_OK;
}

extern "C" int dlesm_halo_plan_peer_connect_rccl(dlesm_halo_plan *p, int nfields)
{
    DLESM_REQUIRE(p != nullptr, "null plan");
    DLESM_REQUIRE(g_comm != nullptr && g_size >= 1, "no communicator (dlesm_comm_init)");
    char mine[DLESM_PEER_BLOB_BYTES];
    if (int rc = dlesm_halo_plan_peer_export(p, g_rank, nfields, mine)) return rc;
    char *d_in = nullptr, *d_all = nullptr;
    std::vector<char> all((size_t)g_size * DLESM_PEER_BLOB_BYTES);
    DLESM_HIP_TRY(hipMalloc((void **)&d_in, DLESM_PEER_BLOB_BYTES));
    DLESM_HIP_TRY(hipMalloc((void **)&d_all, all.size()));
    DLESM_HIP_TRY(hipMemcpy(d_in, mine, DLESM_PEER_BLOB_BYTES, hipMemcpyHostToDevice));
    const ncclResult_t r = ncclAllGather(d_in, d_all, DLESM_PEER_BLOB_BYTES, ncclChar, g_comm, nullptr);
    int rc = DLESM_OK;
    if (r != ncclSuccess) rc = fail(DLESM_ERCCL, "ncclAllGather of the peer blobs: %s", ncclGetErrorString(r));
    if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(DLESM_EHIP, "all-gather of the peer blobs failed");
    if (!rc && hipMemcpy(all.data(), d_all, all.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(DLESM_EHIP, "copy of the peer blobs failed");
    (void)hipFree(d_in);
    (void)hipFree(d_all);
    if (rc) return rc;
    return dlesm_halo_plan_peer_connect(p, g_rank, g_size, all.data());
}

extern "C" int dlesm_halo_plan_peer_connected(const dlesm_halo_plan *p) { return p && p->peer_on ? 1 : 0; }

// Mailbox operations of one plan run one after the other: on one stream that is stream order; across streams, an event.
static int peer_order(dlesm_halo_plan *p, hipStream_t s)
{
    if (p->peer_used && p->peer_last_stream != s) {
        DLESM_REQUIRE(!capturing(s), "the plan's previous mailbox operation was issued on another stream: an event from there cannot be "
                      "made part of this capture -- issue the operations before the capture on the capturing stream");
        DLESM_HIP_TRY(hipEventRecord(p->ev_peer, p->peer_last_stream));
        DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_peer, 0));
    }
    p->peer_used = true;
    p->peer_last_stream = s;
    return DLESM_OK;
}

// the receive strips of the step that is pending, as they sit in the mailbox (parity of that step).
// EVERY message of the plan is listed, whatever the mask: a direction the mask disables carries no cells (an empty strip:
// nothing is copied, its halo keeps its contents as in the reference, parallel_comms_mod.f90:1557-1571) but its arrival
// flag is still raised by the sender and still waited for here.  That is what keeps the two mailbox halves safe when
// the masks of successive operations DIFFER (comm1..comm4 exchanges mixed with edges-only Jacobi steps on one plan):
// a rank can only issue operation N+2 -- which overwrites half N&1 in its neighbours' mailboxes -- after ITS operation
// N+1 has seen the flag N+1 of every neighbour, and a neighbour raises that flag only behind its own operation N in
// stream order, i.e. after it has read half N&1.  With flags only on enabled directions a rank whose operation N+1 did
// not receive from a neighbour never waited for it.
static int peer_in_strips(const dlesm_halo_plan *p, unsigned mask, unsigned long long seq, int nf, PeerJob::In *in, int *n)
{
    *n = 0;
    const double *par = p->peer_rx + (seq & 1) * p->peer_par_len;
    for (size_t m = 0; m < p->recvs.size(); m++) {
        const Msg &r = p->recvs[m];
        DLESM_REQUIRE(*n < PeerJob::MAXM, "more than %d receive messages in one peer step", PeerJob::MAXM);
        const bool on = dir_enabled(mask, r.dir);
        in[(*n)++] = PeerJob::In{r.i0, r.j0, on ? r.nx : 0, on ? r.ny : 0, par + (long)nf * p->ragg[m], p->peer_flags + m};
    }
    return DLESM_OK;
}

static int peer_join(dlesm_halo_plan *p, hipStream_t s)
{
    DLESM_REQUIRE(!capturing(s) || p->peer_pending_captured,
                  "a peer-transport step is in flight: call dlesm_halo_plan_join before capturing a graph");
    PeerStrips st{};
    if (int rc = peer_in_strips(p, p->pending_mask, p->peer_seq, 1, st.s, &st.n)) return rc;
    if (int rc = peer_order(p, s)) return rc;      // the step itself may have run on another stream
    double *one[1] = {p->pending_field};
    if (int rc = launch_peer_unpack(st, p->peer_seq, p->peer_seqw, one, 1, p->ld, p->frame_timed_out, s)) return rc;
    p->peer_pending = false;
    p->pending_field = nullptr;
    return DLESM_OK;
}

// The distributed Jacobi step over the mailboxes: ONE launch on the caller's stream (frame workgroups that store into the
// neighbours' mailboxes + the interior sweep), and in the joined form one small launch behind it that waits for the
// arrival flags and copies the strips into the halos.  No side stream, no RCCL kernel, no pack, no event.
static int jacobi5_step_peer(dlesm_halo_plan *p, const double *in, double *out, int ld, int ny, int xstart, int xstop,
                             int ystart, int ystop, hipStream_t s, bool pipelined)
{
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier distributed step gave up waiting for a flag (frame or halo wait timed out)");
    const unsigned mask = tuning("j5_dm_corners", 0) ? DLESM_DIRS_ALL : (DLESM_DIRS_ALL | DLESM_DIRS_NO_DIAGONALS);
    const bool chain = pipelined && p->peer_pending && p->pending_stream == s && p->pending_field == in &&
                       p->pending_mask == mask && tuning("j5_dm_chain", 1);
    if (!chain)
        if (int rc = join_pending(p, s)) return rc;
    if (int rc = peer_order(p, s)) return rc;
    PeerJob job{};
    const unsigned long long seq = p->peer_seq + 1;
    const int fx0 = xstart - 1, fx1 = xstop - 1, fy0 = ystart - 1, fy1 = ystop - 1;
    for (size_t m = 0; m < p->sends.size(); m++) {
        const Msg &sm = p->sends[m];
        DLESM_REQUIRE(job.nout < PeerJob::MAXM, "more than %d send messages in one peer step", PeerJob::MAXM);
        if (!dir_enabled(mask, sm.dir)) {      // no cells, but the flag is raised all the same (peer_in_strips)
            job.out[job.nout++] = PeerJob::Out{0, 0, 0, 0, p->peer_tx[m], p->peer_txflag[m]};
            continue;
        }
        const bool col = sm.nx == 1 && (sm.i0 == fx0 || sm.i0 == fx1) && sm.j0 >= fy0 && sm.j0 + sm.ny - 1 <= fy1;
        const bool row = sm.ny == 1 && (sm.j0 == fy0 || sm.j0 == fy1) && sm.i0 >= fx0 && sm.i0 + sm.nx - 1 <= fx1;
        DLESM_REQUIRE(col || row, "peer transport: send strip (%d:%d,%d:%d) is not part of the one-cell frame of the box "
                      "(plans of halo depth 1 stepped over their internal region only)", sm.i0 + 1, sm.i0 + sm.nx, sm.j0 + 1, sm.j0 + sm.ny);
        job.out[job.nout++] = PeerJob::Out{sm.i0, sm.j0, sm.nx, sm.ny,
                                           p->peer_tx[m] + (seq & 1) * p->peer_tx_par[m] + p->peer_tx_off[m], p->peer_txflag[m]};
    }
    if (chain) {      // the halos of `in` are still in the mailbox: read them there, once they have arrived
        if (int rc = peer_in_strips(p, mask, seq - 1, 1, job.in, &job.nin)) return rc;
        job.wait_seq = seq - 1;
        job.virt = 1;
    }
    job.seq = seq;
    job.seqw = p->peer_seqw;
    job.counter = p->peer_counter;
    job.wait_ticks = remote_wait_ticks();
    job.timed_out = p->frame_timed_out;
    job.fenced = tuning("mailbox_fences", 1);
    // joined form: the join rides in the same launch (a few workgroups that wait for this step's strips and copy them into
    // out's halos) -- dm_peer_join_fused = 0: the separate wait + unpack launch behind the step
    const bool join_inside = !pipelined && tuning("dm_peer_join_fused", 1);
    if (join_inside)
        if (int rc = peer_in_strips(p, mask, seq, 1, job.un, &job.nun)) return rc;
    bool fused = false;
    if (int rc = launch_stencil5_peer(in, out, ld, ny, xstart, xstop, ystart, ystop, job, s, &fused)) return rc;
    if (!fused) {     // arrays or box the tile kernel does not take: the frame workgroups alone, then the plain interior sweep
        if (int rc = launch_stencil5_peer_frame(in, out, ld, ny, xstart, xstop, ystart, ystop, job, s)) return rc;
        if (int rc = launch_stencil5(in, out, ld, ny, xstart + 1, xstop - 1, ystart + 1, ystop - 1, s)) return rc;
    }
    p->peer_seq = seq;
    if (fused && join_inside && job.nun > 0) return DLESM_OK;       // joined already: nothing pending
    p->peer_pending = true;
    p->peer_pending_captured = capturing(s);
    p->pending_field = out;
    p->pending_mask = mask;
    p->pending_stream = s;
    if (!pipelined) return join_pending(p, s);
    return DLESM_OK;
}

static int jacobi5_step_dm_impl(dlesm_halo_plan *p, const double *in, double *out, int ld, int ny, int xstart,
                                int xstop, int ystart, int ystop, hipStream_t s, bool pipelined)
{
    DLESM_REQUIRE(p != nullptr && in != nullptr && out != nullptr, "null pointer");
    DLESM_REQUIRE(p->ld == ld && p->ny == ny, "plan is for %dx%d fields, got %dx%d", p->ld, p->ny, ld, ny);
    if (int rc = ensure_device()) return rc;
    hipStream_t side = side_stream();
    if (int rc = capture_ok(p, s, p->peer_on && (g_mailbox || tuning("dm_peer", 1)))) return rc;
    const bool comms = !p->sends.empty() || !p->recvs.empty();
    if (!comms) // single tile: one launch over the whole box
        return launch_stencil5(in, out, ld, ny, xstart, xstop, ystart, ystop, s);
    if (p->peer_on && (g_mailbox || tuning("dm_peer", 1)))      // connected mailboxes: the frame workgroups are the exchange
        return jacobi5_step_peer(p, in, out, ld, ny, xstart, xstop, ystart, ystop, s, pipelined);
    // The previous step of a pipelined sequence left its exchange in flight.  If this step can take
    // the one-launch form on the same stream, its frame workgroups wait for that exchange on the
    // device (halo_flag) and the caller's stream needs no event wait at all; otherwise join now.
    const bool graph = capturing(s);
    if (graph) pipelined = false;                        // a captured step joins inside the graph
    const bool can_chain = pipelined && p->pending && p->pending_stream == s && p->frame_flag &&
                           tuning("j5_dm_fused", 1) && tuning("j5_dm_chain", 1) && streams_run_concurrently(s);
    if (!can_chain)
        if (int rc = join_pending(p, s)) return rc;
    // The 5-point stencil never reads a corner halo: exchange the four edge directions only
    // (what passing just the needed comm directions does in the reference, pcomms:1557-1571).
    // j5_dm_corners=1 keeps the diagonal messages (halos then equal a full halo_exchange).
    const unsigned mask = tuning("j5_dm_corners", 0) ? DLESM_DIRS_ALL : (DLESM_DIRS_ALL | DLESM_DIRS_NO_DIAGONALS);
    // 1. the frame of `out`: the only cells a neighbour will ask for.  The west/east columns are
    //    strided in memory; the frame kernel drops each of their cells straight into its slot of
    //    the send buffer as well, so no pack launch sits between the frame and the exchange.
    FramePack fp{};
    bool prepacked = tuning("j5_dm_frame_pack", 1) != 0;
    for (const Msg &m : p->sends) {
        if (m.off < 0 || !dir_enabled(mask, m.dir)) continue;
        const bool on_frame = m.nx == 1 && (m.i0 == xstart - 1 || m.i0 == xstop - 1) && m.j0 >= ystart - 1 &&
                              m.j0 + m.ny - 1 <= ystop - 1;
        if (!on_frame || fp.n == FramePack::MAXS) { prepacked = false; break; }
        fp.s[fp.n++] = FramePack::Col{m.i0, m.j0, m.ny, m.off};
    }
    if (prepacked && fp.n) {
        if (int rc = ensure_buffers(p, 1, s)) return rc;
        fp.buf = p->sendbuf;
    } else {
        fp.n = 0;
    }
    // One launch: the frame as the first workgroups of the interior sweep, its completion published
    // through a flag the side stream is parked on -- no frame launch and no event record on the
    // caller's stream (each costs it microseconds of a ~180 us step, scripts/syncbench.hip).
    bool fused = false;
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier distributed step gave up waiting for a flag (frame or halo wait timed out)");
    if (!graph && p->frame_flag && tuning("j5_dm_fused", 1) && streams_run_concurrently(s)) {
        FrameJob job{};
        job.pk = fp;
        job.counter = p->frame_counter;
        job.flag = p->frame_flag;
        job.seq = p->frame_seq + 1;
        job.halo_flag = p->halo_flag;
        job.halo_seq = can_chain ? p->pending_seq : 0;
        job.timed_out = p->frame_timed_out;
        job.halo_wait_ticks = remote_wait_ticks();
        job.acquire = tuning("dm_acquire", 1);
        // chained step: the previous exchange was not unpacked -- `in`'s west/east halo columns are the
        // received strips themselves, read from the receive buffer (the same mask was exchanged)
        const bool virt = can_chain && p->pending_field == in && tuning("j5_dm_lazy_unpack", 1);
        if (virt) {
            bool ok = true;
            for (const Msg &m : p->recvs) {
                if (m.off < 0 || !dir_enabled(p->pending_mask, m.dir)) continue;
                if (m.nx != 1 || job.nh == FramePack::MAXS) { ok = false; break; }
                job.hs[job.nh++] = FrameJob::HaloCol{m.i0, m.j0, m.ny, m.off};
            }
            if (ok) job.halo_buf = p->recvbuf;
            else job.nh = 0;
        }
        if (can_chain && p->pending_field && !job.halo_buf) {       // cannot read them in place: unpack first
            if (int rc = join_pending(p, s)) return rc;
            job.halo_seq = 0;
        }
        if (int rc = launch_stencil5_framed(in, out, ld, ny, xstart, xstop, ystart, ystop, job, s, &fused)) return rc;
        if (fused) {
            p->frame_seq = job.seq;
            p->pending = false;                          // the chained wait (if any) is inside the launch
            p->pending_field = nullptr;
            // pipelined: the received west/east strips stay in the receive buffer (no strided unpack kernel
            // beside the sweep -- measured: two column-copy kernels cost the sweep 3 us, RCCL itself nothing);
            // the next step reads them there, a join unpacks them
            const bool lazy = pipelined && tuning("j5_dm_lazy_unpack", 1);
            if (int rc = launch_frame_flag_wait(p->frame_flag, job.seq, p->frame_timed_out, side)) return rc;
            if (int rc = exchange_on(p, out, mask, side, prepacked, lazy)) return rc;
            if (int rc = launch_flag_set(p->halo_flag, job.seq, side)) return rc;
            if (lazy) { p->pending_field = out; p->pending_mask = mask; }
            DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
        } else if (can_chain) {                          // the arrays do not qualify: fall back to the event join
            if (int rc = join_pending(p, s)) return rc;
        }
    }
    if (!fused) {
        if (int rc = launch_stencil5_frame(in, out, ld, ny, xstart, xstop, ystart, ystop, s, prepacked ? &fp : nullptr))
            return rc;
        DLESM_HIP_TRY(hipEventRecord(p->ev_frame, s));
        // 2. exchange out's frame on the side stream ...
        DLESM_HIP_TRY(hipStreamWaitEvent(side, p->ev_frame, 0));
        if (int rc = exchange_on(p, out, mask, side, prepacked)) return rc;
        DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
        // 3. ... while the interior streams through HBM on the caller's stream
        if (int rc = launch_stencil5(in, out, ld, ny, xstart + 1, xstop - 1, ystart + 1, ystop - 1, s)) return rc;
    }
    // 4. join: the next step reads out's halos.  Pipelined one-launch form: left to the next step's
    //    frame workgroups (device flag) or to dlesm_halo_plan_join, whichever comes first.
    if (pipelined && fused) {
        p->pending = true;
        p->pending_seq = p->frame_seq;
        p->pending_stream = s;
        return DLESM_OK;
    }
    // joined form of the one-launch step: a one-wave kernel on the caller's stream that sleeps on the
    // exchange's completion flag -- a kernel launch costs the stream ~2.5 us, an event wait 7-8
    // (scripts/syncbench.hip); the halos were released at device scope when the exchange's last kernel ended
    if (fused && tuning("dm_flag_join", 1)) return launch_frame_flag_wait(p->halo_flag, p->frame_seq, p->frame_timed_out, s, true);
    DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
    return DLESM_OK;
}

extern "C" int dlesm_jacobi5_step_dm(dlesm_halo_plan *p, const double *in, double *out, int ld, int ny,
                                     int xstart, int xstop, int ystart, int ystop, void *stream)
{
    return jacobi5_step_dm_impl(p, in, out, ld, ny, xstart, xstop, ystart, ystop, (hipStream_t)stream, false);
}

extern "C" int dlesm_jacobi5_step_dm_pipelined(dlesm_halo_plan *p, const double *in, double *out, int ld, int ny,
                                               int xstart, int xstop, int ystart, int ystop, void *stream)
{
    return jacobi5_step_dm_impl(p, in, out, ld, ny, xstart, xstop, ystart, ystop, (hipStream_t)stream, true);
}

// The distributed step of ANY 3x3 weighted kernel: frame of `out` (its west/east columns written
// into the send buffer too), full eight-direction exchange of `out` on the side stream -- corner
// halos are operands of a 9-point stencil -- while the interior sweeps on the caller's stream,
// join.  A coefficient set with zero corner weights exchanges the four edges only.
extern "C" int dlesm_stencil9_step_dm(dlesm_halo_plan *p, const double *in, double *out, const double *coef, int ld,
                                      int ny, int xstart, int xstop, int ystart, int ystop, void *stream)
{
    DLESM_REQUIRE(p != nullptr && in != nullptr && out != nullptr && coef != nullptr, "null pointer");
    DLESM_REQUIRE(p->ld == ld && p->ny == ny, "plan is for %dx%d fields, got %dx%d", p->ld, p->ny, ld, ny);
    if (int rc = ensure_device()) return rc;
    hipStream_t s = (hipStream_t)stream, side = side_stream();
    if (int rc = capture_ok(p, s, p->peer_on && (g_mailbox || (tuning("dm_peer", 1) && tuning("dm_peer_exchange", 1))))) return rc;
    if (int rc = join_pending(p, s)) return rc;
    if (p->sends.empty() && p->recvs.empty()) return launch_stencil9(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, s);
    const bool corners = coef[0] != 0.0 || coef[2] != 0.0 || coef[6] != 0.0 || coef[8] != 0.0;
    const unsigned mask = corners ? DLESM_DIRS_ALL : (DLESM_DIRS_ALL | DLESM_DIRS_NO_DIAGONALS);
    if (p->peer_on && (g_mailbox || (tuning("dm_peer", 1) && tuning("dm_peer_exchange", 1)))) {
        // mailboxes: the whole box, then the two-launch exchange behind it on the same stream (7 us against an RCCL group's 42)
        if (int rc = launch_stencil9(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, s)) return rc;
        return exchange_on(p, out, mask, s);
    }
    FramePack fp{};
    bool prepacked = true;
    for (const Msg &m : p->sends) {
        if (m.off < 0 || !dir_enabled(mask, m.dir)) continue;
        const bool on_frame = m.nx == 1 && (m.i0 == xstart - 1 || m.i0 == xstop - 1) && m.j0 >= ystart - 1 &&
                              m.j0 + m.ny - 1 <= ystop - 1;
        if (!on_frame || fp.n == FramePack::MAXS) { prepacked = false; break; }
        fp.s[fp.n++] = FramePack::Col{m.i0, m.j0, m.ny, m.off};
    }
    if (prepacked && fp.n) {
        if (int rc = ensure_buffers(p, 1, s)) return rc;
        fp.buf = p->sendbuf;
    } else {
        fp.n = 0;
    }
    // one launch (frame workgroups inside the interior sweep, device flag to the side stream), as the Jacobi step
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier distributed step gave up waiting for a flag (frame or halo wait timed out)");
    if (!capturing(s) && p->frame_flag && tuning("s9_dm_fused", 1) && streams_run_concurrently(s)) {
        FrameJob job{};
        job.pk = fp;
        job.counter = p->frame_counter;
        job.flag = p->frame_flag;
        job.seq = p->frame_seq + 1;
        job.timed_out = p->frame_timed_out;
        job.halo_wait_ticks = remote_wait_ticks();
        bool fused = false;
        if (int rc = launch_stencil9_framed(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, job, s, &fused)) return rc;
        if (fused) {
            p->frame_seq = job.seq;
            if (int rc = launch_frame_flag_wait(p->frame_flag, job.seq, p->frame_timed_out, side)) return rc;
            if (int rc = exchange_on(p, out, mask, side, prepacked)) return rc;
            if (int rc = launch_flag_set(p->halo_flag, job.seq, side)) return rc;
            DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
            if (tuning("dm_flag_join", 1)) return launch_frame_flag_wait(p->halo_flag, job.seq, p->frame_timed_out, s, true);
            DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
            return DLESM_OK;
        }
    }
    if (int rc = launch_stencil9_frame(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, s, prepacked ? &fp : nullptr))
        return rc;
    DLESM_HIP_TRY(hipEventRecord(p->ev_frame, s));
    DLESM_HIP_TRY(hipStreamWaitEvent(side, p->ev_frame, 0));
    if (int rc = exchange_on(p, out, mask, side, prepacked)) return rc;
    DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
    if (int rc = launch_stencil9(in, out, coef, ld, ny, xstart + 1, xstop - 1, ystart + 1, ystop - 1, s)) return rc;
    DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
    return DLESM_OK;
}

// Distributed form of the fused steps (temporal blocking across tiles): `in` holds valid
// depth-nsteps halos, the plan was built from depth-nsteps tables (dlesm_map_comms_depth).
// Stage boxes grow towards every side that has a neighbour; the nsteps-deep frame of `out` is
// computed first (thin boxes along the neighbour sides), its exchange runs on the side stream
// behind the interior, and `out` leaves with valid depth-nsteps halos: ONE exchange per nsteps
// time steps, nsteps times deeper.
extern "C" int dlesm_jacobi5_multi_step_dm(dlesm_halo_plan *p, const double *in, double *out, int ld, int ny,
                                           int nsteps, int xstart, int xstop, int ystart, int ystop,
                                           void *stream)
{
    DLESM_REQUIRE(p != nullptr && in != nullptr && out != nullptr, "null pointer");
    DLESM_REQUIRE(p->ld == ld && p->ny == ny, "plan is for %dx%d fields, got %dx%d", p->ld, p->ny, ld, ny);
    DLESM_REQUIRE(nsteps >= 2 && nsteps <= 8, "fused distributed step: nsteps = %d (2..8 supported)", nsteps);
    if (int rc = ensure_device()) return rc;
    hipStream_t s = (hipStream_t)stream, side = side_stream();
    if (int rc = capture_ok(p, s)) return rc;
    if (int rc = join_pending(p, s)) return rc;
    const int T = nsteps;
    int hasW = 0, hasE = 0, hasS = 0, hasN = 0;
    for (const Msg &m : p->recvs) { // a receive filed under Iminus comes from the west neighbour, ...
        const bool xdir = m.dir == DLESM_IMINUS || m.dir == DLESM_IPLUS;
        const bool ydir = m.dir == DLESM_JMINUS || m.dir == DLESM_JPLUS;
        if (!xdir && !ydir) continue;
        DLESM_REQUIRE((xdir ? m.nx : m.ny) == T, "the plan exchanges depth-%d halos, the fused step needs depth %d",
                      xdir ? m.nx : m.ny, T);
        if (m.dir == DLESM_IMINUS) hasW = 1;
        if (m.dir == DLESM_IPLUS) hasE = 1;
        if (m.dir == DLESM_JMINUS) hasS = 1;
        if (m.dir == DLESM_JPLUS) hasN = 1;
    }
    const int exs = xstart - hasW, exe = xstop + hasE, eys = ystart - hasS, eye = ystop + hasN;
    auto box = [&](int xs, int xe, int ys, int ye) {
        return launch_stencil5_multi(in, out, ld, ny, T, xs, xe, ys, ye, exs, exe, eys, eye, hasW, hasE, hasS, hasN, s);
    };
    const bool comms = !p->sends.empty() || !p->recvs.empty();
    if (!comms) return box(xstart, xstop, ystart, ystop);
    const int ix0 = xstart + hasW * T, ix1 = xstop - hasE * T, iy0 = ystart + hasS * T, iy1 = ystop - hasN * T;
    // From 7 steps per launch the T-deep frame costs more as four thin launches than the exchange it would hide:
    // measured in loop-back at 8 steps, whole box then exchange 0.3024 / 1.0175 ms (8192^2 / 16384^2) against
    // 0.3116 / 1.0188 overlapped; at 4 steps the overlapped form wins (0.2340 / 0.8013 against 0.2468 / 0.8338).
    const bool serial = T >= 7;
    if (serial || ix1 < ix0 || iy1 < iy0) { // ... or the tile is all frame: no interior to hide the exchange behind
        if (int rc = box(xstart, xstop, ystart, ystop)) return rc;
        return exchange_on(p, out, DLESM_DIRS_ALL, s);
    }
    // 1. frame: the T-deep strips along the sides that have a neighbour
    if (hasS)
        if (int rc = box(xstart, xstop, ystart, iy0 - 1)) return rc;
    if (hasN)
        if (int rc = box(xstart, xstop, iy1 + 1, ystop)) return rc;
    if (hasW)
        if (int rc = box(xstart, ix0 - 1, iy0, iy1)) return rc;
    if (hasE)
        if (int rc = box(ix1 + 1, xstop, iy0, iy1)) return rc;
    DLESM_HIP_TRY(hipEventRecord(p->ev_frame, s));
    // 2. exchange of out's frame on the side stream ...
    DLESM_HIP_TRY(hipStreamWaitEvent(side, p->ev_frame, 0));
    if (int rc = exchange_on(p, out, DLESM_DIRS_ALL, side)) return rc;
    DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
    // 3. ... behind the interior
    if (int rc = box(ix0, ix1, iy0, iy1)) return rc;
    DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
    return DLESM_OK;
}

// The distributed shallow-water step over the mailboxes (a plan connected for >= 3 fields): ONE launch whose ring
// workgroups store unew, vnew, pnew of every cell a neighbour needs -- rows, columns and corners, the three fields of a
// message one after the other as in the aggregated exchange -- into that neighbour's mailbox and raise its arrival flags;
// behind it one small launch that waits for this rank's own arrival flags and copies the received strips into the halos of
// the three new fields.  No RCCL kernel, no side stream.  (The time-loop entry takes the same route: the wait sits behind
// the whole sweep in stream order, by which time the neighbours' rings have long arrived.)
static int shallow_step_peer(dlesm_halo_plan *p, const dlesm_sw_params *q, int ld, int ny, int xstart, int xstop, int ystart,
                             int ystop, const double *u, const double *v, const double *pf, const double *uold,
                             const double *vold, const double *pold, double *unew, double *vnew, double *pnew, hipStream_t s,
                             const double *smooth_alpha)
{
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier distributed step gave up waiting for a flag (frame or halo wait timed out)");
    if (int rc = join_pending(p, s)) return rc;
    if (int rc = peer_order(p, s)) return rc;
    const unsigned long long seq = p->peer_seq + 1;
    const int fx0 = xstart - 1, fx1 = xstop - 1, fy0 = ystart - 1, fy1 = ystop - 1;
    SwFrameJob job{};
    FramePack3 &fp = job.pk;
    DLESM_REQUIRE(p->sends.size() <= (size_t)FramePack3::MAXS && p->recvs.size() <= (size_t)PeerJob::MAXM,
                  "more than %d messages in one peer step", FramePack3::MAXS);
    for (size_t k = 0; k < p->sends.size(); k++) {
        const Msg &m = p->sends[k];
        const bool in_box = m.i0 >= fx0 && m.i0 + m.nx - 1 <= fx1 && m.j0 >= fy0 && m.j0 + m.ny - 1 <= fy1;
        const bool on_ring = in_box && ((m.ny == 1 && (m.j0 == fy0 || m.j0 == fy1)) || (m.nx == 1 && (m.i0 == fx0 || m.i0 == fx1)));
        DLESM_REQUIRE(on_ring, "peer transport: send strip (%d:%d,%d:%d) is not part of the one-cell ring of the box (plans of "
                      "halo depth 1 stepped over their internal region only)", m.i0 + 1, m.i0 + m.nx, m.j0 + 1, m.j0 + m.ny);
        fp.s[fp.n] = FramePack3::S{m.i0, m.j0, m.nx, m.ny, 0};
        fp.base[fp.n] = p->peer_tx[k] + (seq & 1) * p->peer_tx_par[k] + 3 * p->peer_tx_off[k];
        job.peer_flag[fp.n] = p->peer_txflag[k];
        fp.n++;
    }
    job.npeer = fp.n;
    job.fenced = tuning("mailbox_fences", 1);
    fp.buf = p->peer_rx;                                  // (never used: every strip has its own base)
    job.counter = p->peer_counter;
    job.flag = p->frame_flag;
    job.seq = seq;
    job.seqw = p->peer_seqw;
    job.timed_out = p->frame_timed_out;
    job.halo_wait_ticks = remote_wait_ticks();
    // the join inside the launch: a few workgroups wait for this step's strips and copy them into the halos of the new fields
    const bool join_inside = tuning("dm_peer_join_fused", 1) != 0;
    if (join_inside)
        if (int rc = peer_in_strips(p, DLESM_DIRS_ALL, seq, 3, job.un, &job.nun)) return rc;
    bool fused = false;
    if (fp.n && tuning("sw_dm_fused", 1))
        if (int rc = launch_shallow_framed(*q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                           job, s, &fused, smooth_alpha))
            return rc;
    if (!fused) {     // arrays or boxes the tile kernel does not take: the ring in its own launch, its flags behind it, the interior
        if (int rc = launch_shallow_frame(*q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                          &fp, s, smooth_alpha))
            return rc;
        if (int rc = launch_peer_flags_set(job.peer_flag, job.npeer, seq, p->peer_seqw, p->frame_timed_out, s)) return rc;
        if (xstop - xstart >= 2 && ystop - ystart >= 2) {
            int rc;
            if (smooth_alpha)
                rc = dlesm_shallow_step_smooth_f64(q, *smooth_alpha, ld, ny, xstart + 1, xstop - 1, ystart + 1, ystop - 1, u, v, pf,
                                                   const_cast<double *>(uold), const_cast<double *>(vold), const_cast<double *>(pold),
                                                   unew, vnew, pnew, s);
            else
                rc = dlesm_shallow_step_f64(q, ld, ny, xstart + 1, xstop - 1, ystart + 1, ystop - 1, u, v, pf, uold, vold, pold, unew,
                                            vnew, pnew, s);
            if (rc) return rc;
        }
    }
    p->peer_seq = seq;
    if (fused && join_inside && job.nun > 0) return DLESM_OK;        // joined inside the launch
    PeerStrips st{};
    if (int rc = peer_in_strips(p, DLESM_DIRS_ALL, seq, 3, st.s, &st.n)) return rc;
    double *fields[3] = {unew, vnew, pnew};
    return launch_peer_unpack(st, seq, p->peer_seqw, fields, 3, ld, p->frame_timed_out, s);
}

// Distributed shallow-water step: the one-cell frame of unew/vnew/pnew first (four thin boxes),
// then ONE grouped exchange of the three new fields on the side stream while the interior is
// computed on the caller's stream; join.  The new fields leave with valid depth-1 halos
// (corners included: the 3x3 footprint needs them).
static int shallow_step_dm_impl(dlesm_halo_plan *p, const dlesm_sw_params *q, int ld, int ny, int xstart,
                                int xstop, int ystart, int ystop, const double *u, const double *v,
                                const double *pf, const double *uold, const double *vold,
                                const double *pold, double *unew, double *vnew, double *pnew,
                                hipStream_t s, bool pipelined, const double *smooth_alpha = nullptr)
{
    DLESM_REQUIRE(p != nullptr && q != nullptr, "null pointer");
    DLESM_REQUIRE(p->ld == ld && p->ny == ny, "plan is for %dx%d fields, got %dx%d", p->ld, p->ny, ld, ny);
    if (int rc = ensure_device()) return rc;
    hipStream_t side = side_stream();
    if (int rc = capture_ok(p, s, p->peer_on && p->peer_fcap >= 3 && (g_mailbox || tuning("dm_peer", 1)))) return rc;
    const bool graph = capturing(s);
    if (graph) pipelined = false;
    // time-loop form: a previous pipelined step on this stream left its exchange in flight; the frame
    // workgroups of this launch wait for it on the device (halo flag) instead of the stream (event)
    const bool can_chain = pipelined && p->pending && p->pending_stream == s && !p->pending_field && p->frame_flag &&
                           tuning("sw_dm_frame", 1) && tuning("sw_dm_fused", 1) && tuning("sw_dm_chain", 1) &&
                           streams_run_concurrently(s);
    if (!can_chain)
        if (int rc = join_pending(p, s)) return rc;
    auto box = [&](int xs, int xe, int ys, int ye) {
        if (smooth_alpha)
            return dlesm_shallow_step_smooth_f64(q, *smooth_alpha, ld, ny, xs, xe, ys, ye, u, v, pf, const_cast<double *>(uold),
                                                 const_cast<double *>(vold), const_cast<double *>(pold), unew, vnew, pnew, s);
        return dlesm_shallow_step_f64(q, ld, ny, xs, xe, ys, ye, u, v, pf, uold, vold, pold, unew, vnew, pnew, s);
    };
    const bool comms = !p->sends.empty() || !p->recvs.empty();
    if (!comms) return box(xstart, xstop, ystart, ystop);
    // mailbox mode has no other transport to fall through to (no communicator): say what is wrong instead of an RCCL error
    DLESM_REQUIRE(!g_mailbox || (p->peer_on && p->peer_fcap >= 3),
                  "mailbox mode: the distributed shallow-water step exchanges three fields, this plan's mailboxes have room for %d "
                  "(DLESM_MAILBOX_FIELDS / tuning mailbox_fields, default 3)", p->peer_fcap);
    if (p->peer_on && p->peer_fcap >= 3 && (g_mailbox || tuning("dm_peer", 1)))      // mailboxes connected for three fields
        return shallow_step_peer(p, q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew, s,
                                 smooth_alpha);
    // 1. frame: the one-cell ring of the box, one cell per thread, all four sides, its west/east columns
    //    written straight into the send buffers of the three new fields -- no pack launches.
    //    sw_dm_frame=0: the round-1 form (four thin boxes + pack kernels).
    bool prepacked = false;
    FramePack3 fp{};
    const bool one_frame = tuning("sw_dm_frame", 1) != 0;
    if (one_frame && tuning("dm_aggregate", 1)) {
        // every message of a depth-1 exchange is a piece of the frame ring: the frame cells go straight
        // into their slots of the aggregated send buffer (3 strips per message), no pack launch
        prepacked = true;
        const int fx0 = xstart - 1, fx1 = xstop - 1, fy0 = ystart - 1, fy1 = ystop - 1;
        for (size_t k = 0; k < p->sends.size(); k++) {
            const Msg &m = p->sends[k];
            const bool in_box = m.i0 >= fx0 && m.i0 + m.nx - 1 <= fx1 && m.j0 >= fy0 && m.j0 + m.ny - 1 <= fy1;
            const bool on_ring = in_box && ((m.ny == 1 && (m.j0 == fy0 || m.j0 == fy1)) ||
                                            (m.nx == 1 && (m.i0 == fx0 || m.i0 == fx1)));
            if (!on_ring || fp.n == FramePack3::MAXS) { prepacked = false; break; }
            fp.s[fp.n++] = FramePack3::S{m.i0, m.j0, m.nx, m.ny, 3 * p->sagg[k]};
        }
        if (prepacked && fp.n) {
            if (int rc = ensure_agg(p, 3, s)) return rc;
            fp.buf = p->sendagg;
        } else {
            prepacked = false;
            fp.n = 0;
        }
    }
    double *fields[3] = {unew, vnew, pnew};
    // One launch (as the Jacobi step): the ring as the first workgroups of the interior sweep, a device flag
    // hands it to the exchange on the side stream -- no frame launch, no event record on the caller's stream.
    DLESM_REQUIRE(!p->frame_timed_out || *(volatile int *)p->frame_timed_out == 0,
                  "an earlier distributed step gave up waiting for a flag (frame wait timed out)");
    if (one_frame && !graph && p->frame_flag && tuning("sw_dm_fused", 1) && streams_run_concurrently(s)) {
        SwFrameJob job{};
        job.pk = fp;
        job.counter = p->frame_counter;
        job.flag = p->frame_flag;
        job.seq = p->frame_seq + 1;
        job.halo_flag = p->halo_flag;
        job.halo_seq = can_chain ? p->pending_seq : 0;
        job.timed_out = p->frame_timed_out;
        job.halo_wait_ticks = remote_wait_ticks();
        job.acquire = tuning("dm_acquire", 1);
        bool fused = false;
        if (int rc = launch_shallow_framed(*q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew,
                                           pnew, job, s, &fused, smooth_alpha))
            return rc;
        if (fused) {
            p->frame_seq = job.seq;
            p->pending = false;                          // the chained wait (if any) is inside the launch
            if (int rc = launch_frame_flag_wait(p->frame_flag, job.seq, p->frame_timed_out, side)) return rc;
            if (int rc = exchange_on(p, fields, 3, DLESM_DIRS_ALL, side, prepacked)) return rc;
            if (int rc = launch_flag_set(p->halo_flag, job.seq, side)) return rc;
            DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
            if (pipelined) {                             // joined by the next step's frame workgroups, or by a join
                p->pending = true;
                p->pending_seq = job.seq;
                p->pending_stream = s;
                p->pending_field = nullptr;
                return DLESM_OK;
            }
            if (tuning("dm_flag_join", 1))               // as in the Jacobi step: a flag-wait kernel instead of an event wait
                return launch_frame_flag_wait(p->halo_flag, job.seq, p->frame_timed_out, s, true);
            DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
            return DLESM_OK;
        }
        if (can_chain)                                   // the arrays do not qualify: fall back to the event join
            if (int rc = join_pending(p, s)) return rc;
    } else if (can_chain) {
        if (int rc = join_pending(p, s)) return rc;
    }
    if (one_frame) {
        if (int rc = launch_shallow_frame(*q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew,
                                          pnew, prepacked ? &fp : nullptr, s, smooth_alpha))
            return rc;
    } else {
        if (int rc = box(xstart, xstop, ystart, ystart)) return rc;
        if (ystop > ystart)
            if (int rc = box(xstart, xstop, ystop, ystop)) return rc;
        if (int rc = box(xstart, xstart, ystart + 1, ystop - 1)) return rc;
        if (xstop > xstart)
            if (int rc = box(xstop, xstop, ystart + 1, ystop - 1)) return rc;
    }
    DLESM_HIP_TRY(hipEventRecord(p->ev_frame, s));
    // 2. grouped exchange of the three new fields on the side stream ...
    DLESM_HIP_TRY(hipStreamWaitEvent(side, p->ev_frame, 0));
    if (int rc = exchange_on(p, fields, 3, DLESM_DIRS_ALL, side, prepacked)) return rc;
    DLESM_HIP_TRY(hipEventRecord(p->ev_comm, side));
    // 3. ... behind the interior
    if (int rc = box(xstart + 1, xstop - 1, ystart + 1, ystop - 1)) return rc;
    DLESM_HIP_TRY(hipStreamWaitEvent(s, p->ev_comm, 0));
    return DLESM_OK;
}

extern "C" int dlesm_shallow_step_dm(dlesm_halo_plan *p, const dlesm_sw_params *q, int ld, int ny, int xstart,
                                     int xstop, int ystart, int ystop, const double *u, const double *v,
                                     const double *pf, const double *uold, const double *vold,
                                     const double *pold, double *unew, double *vnew, double *pnew,
                                     void *stream)
{
    return shallow_step_dm_impl(p, q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                (hipStream_t)stream, false);
}

extern "C" int dlesm_shallow_step_dm_pipelined(dlesm_halo_plan *p, const dlesm_sw_params *q, int ld, int ny,
                                               int xstart, int xstop, int ystart, int ystop, const double *u,
                                               const double *v, const double *pf, const double *uold,
                                               const double *vold, const double *pold, double *unew, double *vnew,
                                               double *pnew, void *stream)
{
    return shallow_step_dm_impl(p, q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                (hipStream_t)stream, true);
}

// The distributed step with the Asselin filter of the old level folded in (dlesm_shallow_step_smooth_f64's arithmetic,
// dlesm_shallow_step_dm's exchange): the filtered old level needs no exchange -- the next step reads it at (i, j) only.
extern "C" int dlesm_shallow_step_smooth_dm(dlesm_halo_plan *p, const dlesm_sw_params *q, double alpha, int ld, int ny,
                                            int xstart, int xstop, int ystart, int ystop, const double *u, const double *v,
                                            const double *pf, double *uold, double *vold, double *pold, double *unew,
                                            double *vnew, double *pnew, void *stream)
{
    return shallow_step_dm_impl(p, q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                (hipStream_t)stream, false, &alpha);
}

extern "C" int dlesm_shallow_step_smooth_dm_pipelined(dlesm_halo_plan *p, const dlesm_sw_params *q, double alpha, int ld,
                                                      int ny, int xstart, int xstop, int ystart, int ystop, const double *u,
                                                      const double *v, const double *pf, double *uold, double *vold,
                                                      double *pold, double *unew, double *vnew, double *pnew, void *stream)
{
    return shallow_step_dm_impl(p, q, ld, ny, xstart, xstop, ystart, ystop, u, v, pf, uold, vold, pold, unew, vnew, pnew,
                                (hipStream_t)stream, true, &alpha);
}

extern "C" int dlesm_global_sum_f64(double *value)
{
    DLESM_REQUIRE(value != nullptr, "null pointer");
    if (g_size <= 1) return DLESM_OK; // the stub's no-op (parallel_utils_stub_mod.f90:148-150)
    if (g_mailbox) {                  // eight bytes per rank over the board, summed in rank order on every rank
        std::vector<double> all(g_size);
        if (int rc = dlesm_board_allgather(value, sizeof(double), all.data())) return rc;
        double sum = 0.0;
        for (int r = 0; r < g_size; r++) sum += all[r];
        *value = sum;
        return DLESM_OK;
    }
    if (int rc = ensure_device()) return rc;
    static double *d = nullptr;
    if (!d) DLESM_HIP_TRY(hipMalloc((void **)&d, sizeof(double)));
    hipStream_t s = side_stream();
    DLESM_HIP_TRY(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, s));
    DLESM_NCCL_TRY(ncclAllReduce(d, d, 1, ncclDouble, ncclSum, g_comm, s));
    DLESM_HIP_TRY(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, s));
    DLESM_HIP_TRY(hipStreamSynchronize(s));
    return DLESM_OK;
}

// MPI_Gather in mailbox mode: the root publishes an IPC handle of its receive buffer over the board, every other rank maps
// it and copies its block STRAIGHT into its slot (one device-to-device copy over xGMI), a second round of the board tells
// the root that all blocks have landed.  A root buffer that cannot be exported (not a hipMalloc allocation) takes the
// blocks through the board instead (host memory).
static int gather_mailbox(const double *send, double *recv, int n)
{
    struct Note { hipIpcMemHandle_t h; unsigned long long gen; int ok; int pad; };
    const size_t bytes = (size_t)n * sizeof(double);
    Note mine;
    memset(&mine, 0, sizeof mine);
    if (g_rank == 0 && n > 0 && !tuning("mailbox_gather_host", 0)) {
        const size_t need = bytes * (size_t)g_size;
        bool ok = true;
        if (g_gather_cap < need) {           // grow: a new allocation, a new handle, a new generation
            if (g_gather_buf) (void)hipFree(g_gather_buf);      // (the others' mappings keep the old pages alive until they close them)
            g_gather_buf = nullptr;
            g_gather_cap = 0;
            ok = hipMalloc(&g_gather_buf, need) == hipSuccess && hipIpcGetMemHandle(&g_gather_handle, g_gather_buf) == hipSuccess;
            if (ok) { g_gather_cap = need; g_gather_gen++; }
            else {
                (void)hipGetLastError();
                if (g_gather_buf) (void)hipFree(g_gather_buf);
                g_gather_buf = nullptr;
            }
        }
        if (ok) { mine.h = g_gather_handle; mine.gen = g_gather_gen; mine.ok = 1; }
    }
    std::vector<Note> notes(g_size);
    if (int rc = dlesm_board_allgather(&mine, sizeof mine, notes.data())) return rc;
    if (n > 0 && notes[0].ok) {
        char why[256] = "";
        if (g_rank != 0) {
            hipError_t e = hipSuccess;
            if (g_gather_map_gen != notes[0].gen) {      // first gather of the job, or the root has grown its buffer
                if (g_gather_map) (void)hipIpcCloseMemHandle(g_gather_map);
                g_gather_map = nullptr;
                g_gather_map_gen = 0;
                e = ipc_open(&g_gather_map, notes[0].h);        // importers take turns, see there
                if (e == hipSuccess) g_gather_map_gen = notes[0].gen;
                else snprintf(why, sizeof why, "hipIpcOpenMemHandle of the root's gather buffer: %s (%s)", hipGetErrorName(e), hipGetErrorString(e));
            }
            if (e == hipSuccess) {
                e = hipMemcpy((char *)g_gather_map + (size_t)g_rank * bytes, send, bytes, hipMemcpyDeviceToDevice);
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (e != hipSuccess) snprintf(why, sizeof why, "copy into the root's gather buffer: %s", hipGetErrorString(e));
            }
            if (why[0]) (void)hipGetLastError();
        }
        char failed = why[0] ? 1 : 0;
        std::vector<char> every(g_size);
        if (int rc2 = dlesm_board_allgather(&failed, 1, every.data())) return rc2;      // all blocks have landed (or not)
        bool any = false;
        for (int r = 0; r < g_size; r++) any |= every[r] != 0;
        if (!any) {
            if (g_rank == 0) {               // own block straight from `send`, the others' from the gather buffer
                DLESM_HIP_TRY(hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice));
                if (g_size > 1)
                    DLESM_HIP_TRY(hipMemcpy((char *)recv + bytes, (char *)g_gather_buf + bytes, bytes * (size_t)(g_size - 1), hipMemcpyDeviceToDevice));
                DLESM_HIP_TRY(hipDeviceSynchronize());      // the buffer is free for the next gather's blocks once this has returned
            }
            return DLESM_OK;
        }
        // some rank could not use the mapping even after its retries (each of them logged and counted, ipc_open): every rank
        // sees the same verdict, so all of them take the route through host memory below -- slower, same result, and LOUD:
        // the fall-back is counted with the retries, which the tests require to be zero
        g_ipc_retries++;
        fprintf(stderr, "dlesm gather: rank %d FALLS BACK to host memory%s%s\n", g_rank, why[0] ? ": " : " (another rank could not map the root's buffer)", why);
    }
    // through host memory
    std::vector<double> block((size_t)n), all;
    if (n > 0 && hipMemcpy(block.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(DLESM_EHIP, "gather: D2H failed");
    if (g_rank == 0) all.resize((size_t)n * g_size);
    if (int rc2 = dlesm_board_allgather(block.data(), bytes, g_rank == 0 ? all.data() : nullptr)) return rc2;
    if (g_rank == 0 && n > 0 && hipMemcpy(recv, all.data(), bytes * g_size, hipMemcpyHostToDevice) != hipSuccess)
        return fail(DLESM_EHIP, "gather: H2D failed");
    return DLESM_OK;
}

extern "C" int dlesm_gather_f64(const double *send, double *recv, int n)
{
    DLESM_REQUIRE(send != nullptr && n >= 0, "bad arguments");
    if (int rc = ensure_device()) return rc;
    hipStream_t s = side_stream();
    if (g_size <= 1) { // the stub's copy (parallel_utils_stub_mod.f90:154-161)
        DLESM_REQUIRE(recv != nullptr, "null receive buffer");
        DLESM_HIP_TRY(hipMemcpyAsync(recv, send, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        DLESM_HIP_TRY(hipStreamSynchronize(s));
        return DLESM_OK;
    }
    // MPI_Gather to root 0 (parallel_utils_mod.f90:242-255) as grouped send/recv
    DLESM_REQUIRE(g_rank != 0 || recv != nullptr, "null receive buffer on root");
    if (g_mailbox) return gather_mailbox(send, recv, n);
    DLESM_REQUIRE(g_comm != nullptr, "gather before dlesm_comm_init");
    DLESM_NCCL_TRY(ncclGroupStart());
    ncclResult_t err = ncclSuccess;
    if (g_rank == 0) {
        for (int r = 0; r < g_size; r++)
            DLESM_NCCL_IN_GROUP(err, ncclRecv(recv + (size_t)r * n, (size_t)n, ncclDouble, r, g_comm, s));
    }
    DLESM_NCCL_IN_GROUP(err, ncclSend(send, (size_t)n, ncclDouble, 0, g_comm, s));
    if (int rc = group_end(err, "gather (ncclSend/ncclRecv)")) return rc;
    DLESM_HIP_TRY(hipStreamSynchronize(s));
    return DLESM_OK;
}

// ---------------------------------------------------------------------------
// Device-side gather / scatter of whole fields (field_mod.f90:1313-1390, 378-389)

// rows dealt round-robin to the workgroups, lanes along the row: no per-element division
__global__ void pack_inner_k(const double *__restrict__ f, int ld, int x0, int y0, int nx, int h, long slot,
                             double *__restrict__ send)
{
    for (int j = blockIdx.x; j < h; j += gridDim.x) {
        const double *src = f + (size_t)(y0 + j) * ld + x0;
        double *dst = send + (size_t)j * nx;
        for (int i = threadIdx.x; i < nx; i += blockDim.x) dst[i] = src[i];
    }
    // the rest of the slot (tiles are uneven): zeroed
    for (long t = (long)nx * h + (long)blockIdx.x * blockDim.x + threadIdx.x; t < slot; t += (long)gridDim.x * blockDim.x)
        send[t] = 0.0;
}

struct GBox { int x0, y0, w, h; };            // 0-based origin in the global array, extent

// Round 3: both copies as ROW SEGMENTS (dlesm_device.h): pairs anchored on the 128-byte lines of the DESTINATION row (the
// source is read with 16-byte loads at whatever 8-byte alignment it has), all four loads of a lane issued before its
// first store, work items numbered row-major so that workgroups sweep both arrays front to back.  The forms above (a
// workgroup per row, 8-byte lanes: 66 % of the HBM peak at 16384^2) remain for unaligned bases and tiny rows.
// Segments of 256 pairs (one per thread) measured 72-73 % for both copies at 16384^2 against 68 % with 1024
// (scripts/segp_probe.py); the read-only checksum is the other way round (81 % with 1024, 60 % with 256).
typedef rs_d2 d2v;

// dst[d0 .. d0+n) <- src[s0 .. s0+n) (element offsets from the array bases), segment sg of that row; n >= ROWSEG_MIN_NX
__device__ __forceinline__ void copy_row_segment(const double *__restrict__ src, long s0, double *__restrict__ dst, long d0,
                                                 int n, int sg, int segp, bool nt)
{
    const long e0 = d0, e1 = d0 + n - 1, shift = s0 - d0;
    const long safe = (e0 + 1) & ~1L;                    // a destination pair wholly inside the row
    RowPair pr[4];
    d2v v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pr[k] = rowseg_pair(e0, e1, sg, segp, threadIdx.x, k);
        v[k] = *(const rs_d2a8 *)(src + shift + (pr[k].full() ? pr[k].el : safe));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = d2v{pin_here(v[k].x), pin_here(v[k].y)};     // all four loads before the first store
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (pr[k].full()) {
            if (nt) __builtin_nontemporal_store(v[k], (d2v *)(dst + pr[k].el));
            else *(d2v *)(dst + pr[k].el) = v[k];
        } else if (pr[k].m0) dst[pr[k].el] = src[shift + pr[k].el];
        else if (pr[k].m1) dst[pr[k].el + 1] = src[shift + pr[k].el + 1];
    }
}

__global__ __launch_bounds__(256) void pack_inner_rowseg(const double *__restrict__ f, int ld, int x0, int y0, int nx, int h,
                                                         int segs, int segp, long slot, double *__restrict__ send, bool nt)
{
    const long items = (long)segs * h;
    if ((long)blockIdx.x < items) {
        const int j = blockIdx.x / segs, sg = blockIdx.x - j * segs;
        copy_row_segment(f, (long)(y0 + j) * ld + x0, send, (long)j * nx, nx, sg, segp, nt);
        return;
    }
    // the rest of the slot (tiles are uneven): zeroed by the workgroups behind the copy
    const long t = (long)nx * h + ((long)blockIdx.x - items) * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (t + 256 * k < slot) send[t + 256 * k] = 0.0;
}

// grid.x = segments of the WIDEST box x rows of the tallest one, grid.y = rank: narrower / shorter boxes leave early
__global__ __launch_bounds__(256) void unpack_gathered_rowseg(const double *__restrict__ recv, long slot,
                                                              const GBox *__restrict__ boxes, int gnx, int segs, int segp,
                                                              double *__restrict__ global, bool nt)
{
    const GBox b = boxes[blockIdx.y];
    const int j = blockIdx.x / segs, sg = blockIdx.x - j * segs;
    if (j >= b.h || b.w < ROWSEG_MIN_NX) return;             // (boxes narrower than that: see the host side)
    copy_row_segment(recv, (long)blockIdx.y * slot + (long)j * b.w, global, (long)(b.y0 + j) * gnx + b.x0, b.w, sg, segp, nt);
}

// grid.y = rank; j outer / i inner as field_mod.f90:1376-1386
__global__ void unpack_gathered_k(const double *__restrict__ recv, long slot, const GBox *__restrict__ boxes,
                                  int gnx, double *__restrict__ global)
{
    const GBox b = boxes[blockIdx.y];
    const double *src = recv + (size_t)blockIdx.y * slot;
    for (int j = blockIdx.x; j < b.h; j += gridDim.x) {
        double *dst = global + (size_t)(b.y0 + j) * gnx + b.x0;
        const double *row = src + (size_t)j * b.w;
        for (int i = threadIdx.x; i < b.w; i += blockDim.x) dst[i] = row[i];
    }
}

// End of round 3: the DESTINATION of the pack is one dense block, and so is the destination of the unpack of a box as wide
// as the global domain (one rank; 1 x Q meshes).  A sweep that is linear in the destination -- thread t <-> destination pair
// t, 256-thread workgroups front to back, exactly the shape of the copy that sets the ceiling -- keeps every store
// workgroup full and 4 KiB-aligned; the source is read at whatever 8-byte alignment it has, rows one pitch apart (an even
// row length: no pair straddles two rows).  util_gather_linear = 0 restores the row segments.
template <bool NTS>
__global__ __launch_bounds__(256) void pack_inner_linear_k(const double *__restrict__ f, int ld, int x0, int y0, int nx, size_t n2,
                                                           size_t slot, double *__restrict__ send)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // destination pair
    if (t < n2) {
        const size_t e = 2 * t, j = e / (unsigned)nx;
        const int i = (int)(e - j * (unsigned)nx);
        const d2v v = *(const rs_d2a8 *)(f + (size_t)(y0 + (long)j) * ld + x0 + i);
        if (NTS) __builtin_nontemporal_store(v, (d2v *)send + t);
        else ((d2v *)send)[t] = v;
    } else {                                                              // the rest of the slot (tiles are uneven): zeroed
        if (2 * t < slot) send[2 * t] = 0.0;
        if (2 * t + 1 < slot) send[2 * t + 1] = 0.0;
    }
}

// The pack of a box that is not as wide as its rows ran at 71-73 % of the HBM peak however the DESTINATION was indexed (row
// segments, or linearly, with the source pairs loaded at their 8-byte alignment).  This form reads the rows WHOLE -- linear in the
// SOURCE, thread t <-> source pair t of rows y0 .. y0+h-1 with their padding, ALIGNED 16-byte loads front to back -- and writes
// only what lies in the box: 76.8 % at 16384^2.  A box that starts at an odd element (internal%xstart = 2, the usual case)
// makes destination pair (e, e+1) the UPPER half of one source pair and the LOWER half of the next: one wave shift (DPP) brings
// it over, lane 63 fetches it (an L2 hit), and the stores stay 16 bytes wide and aligned (even box width).
template <bool ODD, bool NTS>
__global__ __launch_bounds__(256) void pack_inner_srclinear_k(const double *__restrict__ f, int ld, int x0, int y0, int nx, int h,
                                                              unsigned copy_blocks, size_t slot, double *__restrict__ send)
{
    if (blockIdx.x >= copy_blocks) {                         // the rest of the slot (tiles are uneven): zeroed
        const size_t z = (size_t)nx * h + ((size_t)(blockIdx.x - copy_blocks) * 256 + threadIdx.x) * 2;
        if (z < slot) send[z] = 0.0;
        if (z + 1 < slot) send[z + 1] = 0.0;
        return;
    }
    const unsigned hp = (unsigned)ld / 2;                    // pairs per row (>= 256: a workgroup touches at most two rows)
    const size_t first = (size_t)blockIdx.x * 256;
    size_t j = first / hp;
    unsigned pp = (unsigned)(first - j * hp) + threadIdx.x;
    if (pp >= hp) pp -= hp, j++;
    const bool live = j < (size_t)h;
    if (!live) j = (size_t)h - 1;                            // (every lane of a wave takes part in the shift below)
    const double *row = f + ((size_t)y0 + j) * ld;
    const int e = 2 * (int)pp, x1 = x0 + nx - 1;
    const d2v v = *(const d2v *)(row + e);
    double lo, hi;
    int el;                                                   // element of the row that `lo` is
    if (ODD) {
        double nxt = from_upper<true>(v.x);                  // lane + 1's lower element = element e + 2 of this row
        if ((threadIdx.x & 63) == 63) nxt = pp + 1 < hp ? row[e + 2] : 0.0;
        lo = v.y, hi = nxt, el = e + 1;
    } else {
        lo = v.x, hi = v.y, el = e;
    }
    const bool m0 = live && el >= x0 && el <= x1, m1 = live && el + 1 >= x0 && el + 1 <= x1;
    double *dst = send + j * (size_t)nx + (el - x0);
    if (m0 && m1) {
        if (NTS) __builtin_nontemporal_store(d2v{lo, hi}, (d2v *)dst);
        else *(d2v *)dst = d2v{lo, hi};
    } else {
        if (m0) dst[0] = lo;
        if (m1) dst[1] = hi;
    }
}

// n2 pairs from src to dst, both 16-byte aligned: a received box that is as wide as the global array
template <bool NTS>
__global__ __launch_bounds__(256) void copy_pairs_k(const double *__restrict__ src, double *__restrict__ dst, size_t n2)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2) return;
    if (NTS) __builtin_nontemporal_store(((const d2v *)src)[t], (d2v *)dst + t);
    else ((d2v *)dst)[t] = ((const d2v *)src)[t];
}

extern "C" int dlesm_pack_inner_f64(const double *field, int ld, int ny, int xstart, int xstop, int ystart,
                                    int ystop, double *send, long slot, void *stream)
{
    DLESM_REQUIRE(field != nullptr && send != nullptr, "null pointer");
    if (int rc = ensure_device()) return rc;
    if (int rc = check_box("dlesm_pack_inner_f64", ld, ny, xstart, xstop, ystart, ystop, 0)) return rc;
    const int nx = xstop - xstart + 1, h = ystop - ystart + 1;
    const long n = nx > 0 && h > 0 ? (long)nx * h : 0;
    DLESM_REQUIRE(slot >= n, "slot of %ld doubles for a %dx%d region", slot, nx, h);
    if (slot == 0) return DLESM_OK;
    const int glin = tuning("util_rowseg", 1) ? tuning("util_gather_linear", 1) : 0;     // 1: linear in the source (whole rows read), 2: in the destination
    if (glin == 1 && n > 0 && nx % 2 == 0 && ld % 2 == 0 && ld >= 512 && nx >= ld / 2 && (uintptr_t)field % 16 == 0 &&
        (uintptr_t)send % 16 == 0) {
        const size_t copy_blocks = ((size_t)(ld / 2) * h + 255) / 256, zero_blocks = ((size_t)(slot - n) + 511) / 512;
        if (copy_blocks + zero_blocks < ((size_t)1 << 31)) {
            const bool odd = (xstart - 1) % 2 != 0, nt = nt_stores_for(nx, 0, h - 1) != 0;
#define DLESM_PACKSL(OO, NN) hipLaunchKernelGGL((pack_inner_srclinear_k<OO, NN>), dim3((unsigned)(copy_blocks + zero_blocks)), dim3(256), 0, \
                                                (hipStream_t)stream, field, ld, xstart - 1, ystart - 1, nx, h, (unsigned)copy_blocks, (size_t)slot, send)
            if (odd && nt) DLESM_PACKSL(true, true);
            else if (odd) DLESM_PACKSL(true, false);
            else if (nt) DLESM_PACKSL(false, true);
            else DLESM_PACKSL(false, false);
#undef DLESM_PACKSL
            DLESM_HIP_TRY(hipGetLastError());
            return DLESM_OK;
        }
    }
    if (glin == 2 && n > 0 && nx % 2 == 0 && (uintptr_t)field % 8 == 0 && (uintptr_t)send % 16 == 0 &&
        ((size_t)slot + 1) / 2 < ((size_t)1 << 31) * 256) {
        const size_t pairs = ((size_t)slot + 1) / 2;
        if (nt_stores_for(nx, 0, h - 1))
            hipLaunchKernelGGL(pack_inner_linear_k<true>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field,
                               ld, xstart - 1, ystart - 1, nx, (size_t)n / 2, (size_t)slot, send);
        else
            hipLaunchKernelGGL(pack_inner_linear_k<false>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field,
                               ld, xstart - 1, ystart - 1, nx, (size_t)n / 2, (size_t)slot, send);
        DLESM_HIP_TRY(hipGetLastError());
        return DLESM_OK;
    }
    if ((uintptr_t)field % 8 == 0 && (uintptr_t)send % 16 == 0 && (n == 0 || nx >= ROWSEG_MIN_NX) && tuning("util_rowseg", 1)) {
        int segs = 1, segp = 64;
        if (n > 0) rowseg_split(nx, tuning("util_segp", 256), &segs, &segp);
        const long items = n > 0 ? (long)segs * h : 0, zero_blocks = (slot - n + 1023) / 1024;
        if (items + zero_blocks < (1L << 31)) {
            hipLaunchKernelGGL(pack_inner_rowseg, dim3((unsigned)(items + zero_blocks)), dim3(256), 0, (hipStream_t)stream, field,
                               ld, xstart - 1, ystart - 1, n > 0 ? nx : 0, n > 0 ? h : 0, segs, segp, slot, send,
                               nt_stores_for(nx, 0, h - 1) != 0);
            DLESM_HIP_TRY(hipGetLastError());
            return DLESM_OK;
        }
    }
    long blocks = n > 0 ? h : (slot + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_inner_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, ld, xstart - 1,
                       ystart - 1, n > 0 ? nx : 0, n > 0 ? h : 0, slot, send);
    DLESM_HIP_TRY(hipGetLastError());
    return DLESM_OK;
}

extern "C" int dlesm_unpack_gathered_f64(const double *recv, long slot, const dlesm_decomp *d,
                                         const dlesm_subdomain *subs, int nranks, double *global, void *stream)
{
    DLESM_REQUIRE(recv != nullptr && d != nullptr && subs != nullptr && global != nullptr, "null pointer");
    DLESM_REQUIRE(nranks >= 1 && nranks <= d->ndomains, "%d ranks for %d subdomains", nranks, d->ndomains);
    if (int rc = ensure_device()) return rc;
    std::vector<GBox> boxes(nranks);
    long widest = 0;
    int tallest = 0;
    for (int r = 0; r < nranks; r++) {
        const dlesm_region &g = subs[r].global;
        const int w = g.xstop - g.xstart + 1, h = g.ystop - g.ystart + 1;
        DLESM_REQUIRE(g.xstart >= 1 && g.ystart >= 1 && g.xstop <= d->global_nx && g.ystop <= d->global_ny && w >= 0 &&
                          h >= 0 && (long)w * h <= slot,
                      "subdomain %d box (%d:%d,%d:%d) does not fit (global %dx%d, slot %ld)", r + 1, g.xstart, g.xstop,
                      g.ystart, g.ystop, d->global_nx, d->global_ny, slot);
        boxes[r] = GBox{g.xstart - 1, g.ystart - 1, w, h};
        if ((long)w * h > widest) widest = (long)w * h;
        if (h > tallest) tallest = h;
    }
    if (widest == 0) return DLESM_OK;
    hipStream_t s = (hipStream_t)stream;
    {   // every box as wide as the global array (one rank, 1 x Q meshes): each is one contiguous block on both sides
        bool linear = tuning("util_rowseg", 1) && tuning("util_gather_linear", 1) && (uintptr_t)recv % 16 == 0 && (uintptr_t)global % 16 == 0;
        for (int r = 0; r < nranks && linear; r++) {
            const GBox &b = boxes[r];
            if ((long)b.w * b.h == 0) continue;
            linear = b.w == d->global_nx && b.x0 == 0 && ((long)b.w * b.h) % 2 == 0 && ((long)r * slot) % 2 == 0 &&
                     ((long)b.y0 * d->global_nx) % 2 == 0;
        }
        if (linear) {
            const bool nt = nt_stores_for(d->global_nx, 0, d->global_ny - 1) != 0;
            for (int r = 0; r < nranks; r++) {
                const GBox &b = boxes[r];
                const size_t n2 = (size_t)b.w * b.h / 2;
                if (n2 == 0) continue;
                const double *src = recv + (size_t)r * slot;
                double *dst = global + (size_t)b.y0 * d->global_nx;
                if (nt) hipLaunchKernelGGL(copy_pairs_k<true>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, src, dst, n2);
                else hipLaunchKernelGGL(copy_pairs_k<false>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, src, dst, n2);
            }
            DLESM_HIP_TRY(hipGetLastError());
            DLESM_HIP_TRY(hipStreamSynchronize(s));       // (the entry returns with the copy done, as the table form below does)
            return DLESM_OK;
        }
    }
    GBox *dboxes = nullptr;
    DLESM_HIP_TRY(hipMalloc((void **)&dboxes, boxes.size() * sizeof(GBox)));
    hipError_t e = hipMemcpyAsync(dboxes, boxes.data(), boxes.size() * sizeof(GBox), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        int widest_w = 0;
        for (const GBox &b : boxes) widest_w = b.w > widest_w ? b.w : widest_w;
        int segs, segp, narrowest_w = widest_w;
        for (const GBox &b : boxes)
            if (b.w > 0 && b.h > 0 && b.w < narrowest_w) narrowest_w = b.w;
        rowseg_split(widest_w, tuning("util_segp", 256), &segs, &segp);
        if ((uintptr_t)recv % 8 == 0 && (uintptr_t)global % 16 == 0 && narrowest_w >= ROWSEG_MIN_NX &&
            (long)segs * tallest < (1L << 31) && tuning("util_rowseg", 1)) {
            hipLaunchKernelGGL(unpack_gathered_rowseg, dim3((unsigned)((long)segs * tallest), (unsigned)nranks), dim3(256), 0, s, recv,
                               slot, dboxes, d->global_nx, segs, segp, global, nt_stores_for(d->global_nx, 0, d->global_ny - 1) != 0);
        } else {
            long gx = tallest;
            if (gx > 2048) gx = 2048;
            hipLaunchKernelGGL(unpack_gathered_k, dim3((unsigned)gx, (unsigned)nranks), dim3(256), 0, s, recv, slot, dboxes,
                               d->global_nx, global);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);      // the table is a stack-lifetime upload
    (void)hipFree(dboxes);
    if (e != hipSuccess) return fail(DLESM_EHIP, "unpack of the gathered slots failed: %s", hipGetErrorString(e));
    return DLESM_OK;
}

extern "C" int dlesm_gather_inner_f64(const double *field, int ld, int ny, const dlesm_region *it,
                                      const dlesm_decomp *d, const dlesm_subdomain *subs, int nranks,
                                      double *global_host)
{
    DLESM_REQUIRE(field != nullptr && it != nullptr && d != nullptr && subs != nullptr, "null pointer");
    DLESM_REQUIRE(nranks >= 1, "nranks = %d", nranks);
    if (int rc = ensure_device()) return rc;
    hipStream_t s = side_stream();
    DLESM_HIP_TRY(hipDeviceSynchronize());                 // the field's producers run on the caller's streams
    const bool root = nranks == 1 || g_rank == 0;
    DLESM_REQUIRE(!root || global_host != nullptr, "null result array on the gathering rank");
    const size_t gbytes = (size_t)d->global_nx * d->global_ny * sizeof(double);
    if (nranks == 1) {
        // the copy-out of field_mod.f90:1332-1343: WHATEVER internal region the field has goes to the top-left corner of
        // global_data -- for SW-offset fields with external boundaries on U / V / F points (xstart + 1 / ystart + 1,
        // field_mod.f90:724, 842, 1044, 1055) and for GO_ALL_POINTS fields that is not the whole domain; the rest of
        // global_data keeps what the caller put there (the reference leaves it unset)
        const int w = it->xstop - it->xstart + 1, h = it->ystop - it->ystart + 1;
        if (w <= 0 || h <= 0) return DLESM_OK;
        DLESM_REQUIRE(w <= d->global_nx && h <= d->global_ny, "one rank: internal region %dx%d is larger than the %dx%d domain", w,
                      h, d->global_nx, d->global_ny);
        if (int rc = check_box("dlesm_gather_inner_f64", ld, ny, it->xstart, it->xstop, it->ystart, it->ystop, 0)) return rc;
        DLESM_HIP_TRY(hipMemcpy2D(global_host, (size_t)d->global_nx * sizeof(double),
                                  field + lin(ld, it->xstart, it->ystart), (size_t)ld * sizeof(double),
                                  (size_t)w * sizeof(double), (size_t)h, hipMemcpyDeviceToHost));
        return DLESM_OK;
    }
    DLESM_REQUIRE((g_comm != nullptr || g_mailbox) && g_size == nranks, "gather over %d ranks, communicator has %d", nranks, g_size);
    const int halo_x = it->xstart - 1, halo_y = it->ystart - 1;                  // field_mod.f90:1348-1349
    const long slot = (long)(d->max_width - 2 * halo_x) * (d->max_height - 2 * halo_y);
    DLESM_REQUIRE(slot > 0, "empty gather slot (max tile %dx%d, halos %d,%d)", d->max_width, d->max_height, halo_x, halo_y);
    double *send = nullptr, *recv = nullptr, *gdev = nullptr;
    int rc = DLESM_OK;
    auto cleanup = [&]() {
        if (send) (void)hipFree(send);
        if (recv) (void)hipFree(recv);
        if (gdev) (void)hipFree(gdev);
    };
    if (hipMalloc((void **)&send, (size_t)slot * sizeof(double)) != hipSuccess ||
        (root && (hipMalloc((void **)&recv, (size_t)slot * nranks * sizeof(double)) != hipSuccess ||
                  hipMalloc((void **)&gdev, gbytes) != hipSuccess))) {
        cleanup();
        return fail(DLESM_EHIP, "gather_inner_data: device buffers");
    }
    rc = dlesm_pack_inner_f64(field, ld, ny, it->xstart, it->xstop, it->ystart, it->ystop, send, slot, s);
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(DLESM_EHIP, "gather_inner_data: pack failed");
    if (!rc) rc = dlesm_gather_f64(send, recv, (int)slot);
    if (!rc && root) {
        rc = dlesm_unpack_gathered_f64(recv, slot, d, subs, nranks, gdev, s);
        if (!rc && hipMemcpy(global_host, gdev, gbytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(DLESM_EHIP, "gather_inner_data: copy of the global array to the host failed");
    }
    cleanup();
    return rc;
}

extern "C" int dlesm_scatter_inner_f64(const double *global_host, int gnx, int gny, const dlesm_subdomain *sub,
                                       double *field, int ld, int ny)
{
    DLESM_REQUIRE(global_host != nullptr && sub != nullptr && field != nullptr, "null pointer");
    if (int rc = ensure_device()) return rc;
    const dlesm_region &g = sub->global, &it = sub->internal;
    const int w = it.xstop - it.xstart + 1, h = it.ystop - it.ystart + 1;
    if (w <= 0 || h <= 0) return DLESM_OK;
    DLESM_REQUIRE(g.xstart >= 1 && g.ystart >= 1 && g.xstart + w - 1 <= gnx && g.ystart + h - 1 <= gny,
                  "patch (%d,%d)+%dx%d outside the %dx%d global array", g.xstart, g.ystart, w, h, gnx, gny);
    if (int rc = check_box("dlesm_scatter_inner_f64", ld, ny, it.xstart, it.xstop, it.ystart, it.ystop, 0)) return rc;
    DLESM_HIP_TRY(hipMemcpy2D(field + lin(ld, it.xstart, it.ystart), (size_t)ld * sizeof(double),
                              global_host + lin(gnx, g.xstart, g.ystart), (size_t)gnx * sizeof(double),
                              (size_t)w * sizeof(double), (size_t)h, hipMemcpyHostToDevice));
    return DLESM_OK;
}
