// Runtime plumbing of libdlesm_hip.so: error reporting, device binding,
// device-resident field descriptors and the two device-sync callbacks that
// dl_esm_inf's r2d_field already knows how to call (field_mod.f90:65-105).
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "dlesm_internal.h"

namespace dlesm {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void clear_error() { g_err[0] = '\0'; }

static std::mutex g_mu;
static bool g_ready = false;
static int g_device = -1;
static hipStream_t g_side = nullptr, g_xfer = nullptr;
static std::map<std::string, int> g_tuning;

// DLESM_DM_SAFE=1 (environment, read once) or dlesm_set_tuning("dm_safe", 1): ONE switch that takes every distributed
// step to its conservative form -- frame in its own launch, the exchange ordered by events on both sides, every received
// strip unpacked into the field before anything reads it, no device-side flag waits at all.  The form to fall back to
// when a first run on new hardware (several GPUs over xGMI) fails its self-check: it relies on kernel boundaries and
// stream-ordered events only.
static const char *const kSafeOff[] = {"j5_dm_fused", "sw_dm_fused", "s9_dm_fused", "dm_flag_join", "j5_dm_lazy_unpack",
                                       "j5_dm_chain", "sw_dm_chain", "dm_peer"};
static bool dm_safe_nolock()
{
    static const bool env = [] { const char *e = getenv("DLESM_DM_SAFE"); return e && *e && strcmp(e, "0") != 0; }();
    auto it = g_tuning.find("dm_safe");
    return it != g_tuning.end() ? it->second != 0 : env;
}

// Every setting the library reads, by class (include/dlesm_hip.h, dlesm_set_tuning): USER = for host programs, HOOK = forces
// a path the library also takes by itself (tests reach it on any input), LAB = comparison-only kernels and diagnostics
// that exist in libdlesm_hip_lab.so only.  tests/test_cabi_host.py checks that every key the sources read is listed here.
enum { KEY_USER = 0, KEY_HOOK = 1, KEY_LAB = 2 };
static const struct { const char *key; int cls; } kKeys[] = {
    // ---- USER
    {"dm_safe", KEY_USER}, {"dm_wait_seconds", KEY_USER}, {"dm_peer", KEY_USER}, {"dm_peer_exchange", KEY_USER},
    {"mailbox_fences", KEY_USER}, {"mailbox_fields", KEY_USER}, {"mailbox_gather_host", KEY_USER}, {"dm_acquire", KEY_USER},
    {"j5_dm_corners", KEY_USER}, {"j5_nt_stores", KEY_USER}, {"j5_use_tuned", KEY_USER}, {"side_stream_priority", KEY_USER},
    {"dm_graph_force", KEY_USER}, {"mailbox_finegrained", KEY_USER},
    // ---- HOOK: launch shapes the planning calls choose from
    {"j5_tpb", KEY_HOOK}, {"j5_pad_tiles", KEY_HOOK}, {"j5_autoshape", KEY_HOOK}, {"j5_skew", KEY_HOOK}, {"j5_tile_rows", KEY_HOOK}, {"sw_nt", KEY_HOOK}, {"swk_nt", KEY_HOOK}, {"swk_ntl", KEY_HOOK},
    {"cont_nt", KEY_HOOK}, {"sw_smooth_ntl", KEY_HOOK},
    // ---- HOOK: fall-back kernels (unaligned bases, odd pitches, thin boxes) and the forms DLESM_DM_SAFE / a capture select
    {"j5_variant", KEY_HOOK}, {"sw_kernel", KEY_HOOK}, {"swk_kernel", KEY_HOOK}, {"s9_kernel", KEY_HOOK}, {"j5m_kernel", KEY_HOOK},
    {"cont_kernel", KEY_HOOK}, {"sw_wrap_fused", KEY_HOOK}, {"sw_smooth_fused", KEY_HOOK},
    {"sw_x2_fused", KEY_HOOK},
    {"util_rowseg", KEY_HOOK}, {"util_rowlinear", KEY_HOOK}, {"util_gather_linear", KEY_HOOK},
    {"j5_dm_fused", KEY_HOOK}, {"sw_dm_fused", KEY_HOOK}, {"s9_dm_fused", KEY_HOOK}, {"j5_dm_chain", KEY_HOOK}, {"sw_dm_chain", KEY_HOOK},
    {"j5_dm_lazy_unpack", KEY_HOOK}, {"j5_dm_frame_pack", KEY_HOOK}, {"dm_flag_join", KEY_HOOK}, {"dm_aggregate", KEY_HOOK},
    {"dm_aggregate_single", KEY_HOOK}, {"dm_peer_one_launch", KEY_HOOK}, {"dm_peer_join_fused", KEY_HOOK}, {"dm_inject_timeout", KEY_HOOK},
    // (tests only: sw_dm_frame = 0 is the ring as four thin boxes of the plain step, no kernel of its own; dm_skip_parts
    //  switches parts of an RCCL exchange OFF -- results then come from the mailboxes or are wrong by design)
    {"sw_dm_frame", KEY_HOOK}, {"dm_skip_parts", KEY_HOOK},
    // ---- LAB
    {"j5_kernel", KEY_LAB}, {"j5_rows", KEY_LAB}, {"j5_unroll", KEY_LAB}, {"j5_padw", KEY_LAB}, {"j5xt_rows", KEY_LAB}, {"j5xt_dpp", KEY_LAB},
    {"j5xt_march", KEY_LAB}, {"j5xt_march_slots", KEY_LAB}, {"j5xt_march_ring", KEY_LAB}, {"j5xt_march_perm", KEY_LAB},
    {"sw_tile_rows", KEY_LAB}, {"sw_dpp", KEY_LAB}, {"sw_stack", KEY_LAB}, {"sw_dm_diag", KEY_LAB},
    {"dm_event_system_fence", KEY_LAB}, {"util_segp", KEY_LAB},
    {"sw_x2_rows", KEY_LAB}, {"sw_x2_nt", KEY_LAB}, {"sw_x2_stack", KEY_LAB}, {"sw_x2_pad", KEY_LAB}, {"sw_x2_sw_form", KEY_LAB},
};
static int key_class(const char *key)
{
    for (const auto &k : kKeys)
        if (!strcmp(k.key, key)) return k.cls;
    return -1;
}

static int tuning_nolock(const char *key, int fallback)
{
    if (!kLab && key_class(key) == KEY_LAB) return fallback;      // the product library has no such form
    if (dm_safe_nolock())
        for (const char *k : kSafeOff)
            if (!strcmp(k, key)) return 0;
    auto it = g_tuning.find(key);
    return it == g_tuning.end() ? fallback : it->second;
}

// One pinned host word for the whole process, raised (system scope) by any device-side wait that gives up.  Sticky:
// every later device entry point fails loudly (ensure_device) until the host program acknowledges it.
static int *g_wait_timed_out = nullptr;

static int bind_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return fail(DLESM_ENODEV, "no HIP device available (%s)",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) return fail(DLESM_EINVAL, "device %d out of range [0,%d)", device, n);
    DLESM_HIP_TRY(hipSetDevice(device));
    if (!g_side) {
        // the exchange must not queue behind the interior sweep it hides under: highest priority
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        const int prio = tuning_nolock("side_stream_priority", 1) ? hi : 0;
        DLESM_HIP_TRY(hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, prio));
    }
    if (!g_xfer) DLESM_HIP_TRY(hipStreamCreateWithFlags(&g_xfer, hipStreamNonBlocking));
    if (!g_wait_timed_out) {
        DLESM_HIP_TRY(hipHostMalloc((void **)&g_wait_timed_out, sizeof(int), hipHostMallocMapped));
        *g_wait_timed_out = 0;
    }
    g_device = device;
    g_ready = true;
    return DLESM_OK;
}

static int timed_out_error()
{
    return fail(DLESM_EHIP, "a distributed step gave up waiting on a device flag (a frame that never reported, or an exchange "
                            "whose messages did not arrive within dm_wait_seconds): everything enqueued behind that wait may "
                            "have read halos that had not arrived -- results since then are INVALID.  Destroy the halo plans, "
                            "then dlesm_wait_timed_out(1) to acknowledge (or restart the program)");
}

int ensure_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready) return (g_wait_timed_out && *(volatile int *)g_wait_timed_out) ? timed_out_error() : DLESM_OK;
    // lazily adopt the device the host program already selected (e.g. torch.cuda.set_device)
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        cur = 0;
    }
    return bind_device(cur);
}

int *wait_timed_out_word() { return g_wait_timed_out; }
hipStream_t side_stream() { return g_side; }
hipStream_t transfer_stream() { return g_xfer; }

int tuning(const char *key, int fallback)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return tuning_nolock(key, fallback);
}

} // namespace dlesm

using namespace dlesm;

extern "C" const char *dlesm_last_error(void) { return g_err; }
extern "C" int dlesm_version(void) { return DLESM_VERSION; }

extern "C" int dlesm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int dlesm_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready && device == g_device) return DLESM_OK;
    return bind_device(device);
}

extern "C" int dlesm_wait_timed_out(int clear)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_wait_timed_out) return 0;
    const int was = *(volatile int *)g_wait_timed_out != 0;
    if (clear && was) {
        (void)hipDeviceSynchronize();                     // nothing that could still raise it is left in flight
        *(volatile int *)g_wait_timed_out = 0;
        invalidate_concurrency_probe();                   // whatever made the wait fail may have changed how streams run
    }
    return was;
}

extern "C" int dlesm_finalize(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready) return DLESM_OK;
    (void)hipDeviceSynchronize();
    // The time-out word is NOT freed: halo plans cache its address (and hand it to kernels), and a plan may outlive a
    // finalize / re-initialise pair of the host program.  Four bytes of pinned memory for the life of the process; the
    // word is cleared so that a re-initialised library starts clean.
    if (g_wait_timed_out) *(volatile int *)g_wait_timed_out = 0;
    invalidate_concurrency_probe();
    if (g_side) (void)hipStreamDestroy(g_side);
    if (g_xfer) (void)hipStreamDestroy(g_xfer);
    g_side = g_xfer = nullptr;
    g_ready = false;
    return DLESM_OK;
}

extern "C" int dlesm_tuning_class(const char *key) { return key ? key_class(key) : -1; }
extern "C" int dlesm_is_lab_build(void) { return kLab ? 1 : 0; }

extern "C" int dlesm_set_tuning(const char *key, int value)
{
    if (!key) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    const int cls = key_class(key);
    if (cls < 0 || (cls == KEY_LAB && !kLab)) {      // said once per key: a typo, or a lab key in the product library
        static std::map<std::string, bool> said;
        if (!said[key]) {
            said[key] = true;
            if (cls < 0) fprintf(stderr, "dlesm_set_tuning: unknown key \"%s\" (kept; nothing reads it)\n", key);
            else fprintf(stderr, "dlesm_set_tuning: \"%s\" selects a comparison-only form that exists in libdlesm_hip_lab.so only: "
                                 "ignored by this library\n", key);
        }
    }
    int prev = g_tuning.count(key) ? g_tuning[key] : 0;
    g_tuning[key] = value;
    // diagnostic: raise the process-wide time-out word exactly as a device-side wait that gives up does
    if (!strcmp(key, "dm_inject_timeout") && value && g_wait_timed_out) *(volatile int *)g_wait_timed_out = 1;
    return prev;
}

// ---------------------------------------------------------------------------
// field descriptors

static int check_field(const dlesm_field *f)
{
    if (!f || f->magic != DLESM_FIELD_MAGIC) return fail(DLESM_EINVAL, "not a dlesm_field descriptor");
    return DLESM_OK;
}

extern "C" int dlesm_field_create(int ld, int ny, dlesm_field **out)
{
    DLESM_REQUIRE(out != nullptr, "null output pointer");
    DLESM_REQUIRE(ld > 0 && ny > 0, "field extents %dx%d", ld, ny);
    if (int rc = ensure_device()) return rc;
    const size_t bytes = (size_t)ld * (size_t)ny * sizeof(double);
    double *p = nullptr;
    DLESM_HIP_TRY(hipMalloc((void **)&p, bytes));
    // "explicitly set all elements to 0" (field_mod.f90:357-376) -- and the fill has LANDED when this returns.  hipMemset of
    // device memory is enqueued on the null stream and returns at once: behind a kernel that is still running there it ran
    // AFTER the (blocking, transfer-stream) upload of field_to_device and zeroed what had just been uploaded (found in round 4
    // by the Fortran two-step test, whose reference fields go to the device while the kernel under test is running; the
    // Python constructor had the same fault in round 3).  So: on the transfer stream -- the stream every later upload of this
    // field uses -- and waited for, which does not wait for the caller's kernels.
    hipError_t e = hipMemsetAsync(p, 0, bytes, transfer_stream());
    if (e == hipSuccess) e = hipStreamSynchronize(transfer_stream());
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(DLESM_EHIP, "zero-fill of a new field failed: %s", hipGetErrorString(e));
    }
    dlesm_field *f = new dlesm_field{DLESM_FIELD_MAGIC, p, ld, ny, true};
    *out = f;
    return DLESM_OK;
}

extern "C" int dlesm_field_wrap(void *device_data, int ld, int ny, dlesm_field **out)
{
    DLESM_REQUIRE(out != nullptr && device_data != nullptr, "null pointer");
    DLESM_REQUIRE(ld > 0 && ny > 0, "field extents %dx%d", ld, ny);
    *out = new dlesm_field{DLESM_FIELD_MAGIC, (double *)device_data, ld, ny, false};
    return DLESM_OK;
}

extern "C" int dlesm_field_destroy(dlesm_field *f)
{
    if (!f) return DLESM_OK;
    if (int rc = check_field(f)) return rc;
    if (f->owned && f->data) DLESM_HIP_TRY(hipFree(f->data));
    f->magic = 0;
    delete f;
    return DLESM_OK;
}

extern "C" double *dlesm_field_data(const dlesm_field *f) { return check_field(f) ? nullptr : f->data; }
extern "C" int dlesm_field_ld(const dlesm_field *f) { return check_field(f) ? -1 : f->ld; }
extern "C" int dlesm_field_ny(const dlesm_field *f) { return check_field(f) ? -1 : f->ny; }

// ---------------------------------------------------------------------------
// B1: the reference's C-flavour sync callbacks.  Fatal on error, like every
// error path of the reference (gocean_stop -> parallel_abort).

[[noreturn]] static void die(const char *what)
{
    fprintf(stderr, " %s: %s\n", what, g_err);
    abort();
}

static int copy_patch(const dlesm_field *f, double *host, int startx, int starty, int nx, int ny,
                      bool to_device, bool blocking)
{
    if (int rc = check_field(f)) return rc;
    DLESM_REQUIRE(host != nullptr, "null host pointer");
    DLESM_REQUIRE(startx >= 1 && starty >= 1 && nx >= 0 && ny >= 0 && startx + nx - 1 <= f->ld &&
                      starty + ny - 1 <= f->ny,
                  "patch (%d,%d)+%dx%d outside field %dx%d", startx, starty, nx, ny, f->ld, f->ny);
    if (nx == 0 || ny == 0) return DLESM_OK;
    if (int rc = ensure_device()) return rc;
    const size_t off = lin(f->ld, startx, starty);
    const size_t pitch = (size_t)f->ld * sizeof(double);
    hipStream_t s = transfer_stream();
    if (nx == f->ld) {
        // whole rows are one contiguous block
        const size_t bytes = pitch * (size_t)ny;
        if (to_device) DLESM_HIP_TRY(hipMemcpyAsync(f->data + off, host + off, bytes, hipMemcpyHostToDevice, s));
        else DLESM_HIP_TRY(hipMemcpyAsync(host + off, f->data + off, bytes, hipMemcpyDeviceToHost, s));
    } else if (to_device) {
        DLESM_HIP_TRY(hipMemcpy2DAsync(f->data + off, pitch, host + off, pitch, (size_t)nx * sizeof(double),
                                       (size_t)ny, hipMemcpyHostToDevice, s));
    } else {
        DLESM_HIP_TRY(hipMemcpy2DAsync(host + off, pitch, f->data + off, pitch, (size_t)nx * sizeof(double),
                                       (size_t)ny, hipMemcpyDeviceToHost, s));
    }
    if (blocking) DLESM_HIP_TRY(hipStreamSynchronize(s));
    return DLESM_OK;
}

extern "C" void dlesm_read_from_device(void *from, void *to, int startx, int starty, int nx, int ny,
                                       bool blocking)
{
    // the kernels run on the caller's streams: make their results visible first
    if (hipDeviceSynchronize() != hipSuccess) {
        fail(DLESM_EHIP, "hipDeviceSynchronize failed");
        die("dlesm_read_from_device");
    }
    if (copy_patch((const dlesm_field *)from, (double *)to, startx, starty, nx, ny, false, blocking))
        die("dlesm_read_from_device");
}

extern "C" void dlesm_write_to_device(void *from, void *to, int startx, int starty, int nx, int ny,
                                      bool blocking)
{
    // The upload runs on the library's transfer stream, which is ordered against nothing the caller has launched: a kernel
    // that still reads (or writes) this field on one of the caller's streams must have finished before the host copy
    // replaces it -- the reference's device is synchronous (tests/device_computation/test_device_io.f90), so a host program
    // written against it expects exactly that.  Uploads are initialisation / set_data traffic, never inside a time step.
    if (hipDeviceSynchronize() != hipSuccess) {
        fail(DLESM_EHIP, "hipDeviceSynchronize failed");
        die("dlesm_write_to_device");
    }
    if (copy_patch((const dlesm_field *)to, (double *)from, startx, starty, nx, ny, true, blocking))
        die("dlesm_write_to_device");
}

extern "C" int dlesm_transfer_sync(void)
{
    if (!transfer_stream()) return DLESM_OK;
    DLESM_HIP_TRY(hipStreamSynchronize(transfer_stream()));
    return DLESM_OK;
}
