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

static int tuning_nolock(const char *key, int fallback)
{
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

extern "C" int dlesm_set_tuning(const char *key, int value)
{
    if (!key) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
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
    // "explicitly set all elements to 0" (field_mod.f90:357-376)
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(DLESM_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
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
    if (copy_patch((const dlesm_field *)to, (double *)from, startx, starty, nx, ny, true, blocking))
        die("dlesm_write_to_device");
}

extern "C" int dlesm_transfer_sync(void)
{
    if (!transfer_stream()) return DLESM_OK;
    DLESM_HIP_TRY(hipStreamSynchronize(transfer_stream()));
    return DLESM_OK;
}
