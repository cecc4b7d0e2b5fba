// What does each cross-stream dependency of the overlapped distributed step cost on the main
// stream?  One iteration = [small "frame" kernel] + [big streaming "interior" kernel ~180 us] on the
// main stream, a small "exchange" kernel on a side stream, joined in different ways.  Reports us per
// iteration minus the plain big kernel.    hipcc --offload-arch=gfx950 -O3 syncbench.hip -o syncbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void big(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// the same, but block 0 also plays "frame": writes a little and then raises a flag (release, system scope)
__global__ void big_signal(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n, unsigned long long *flag,
                           unsigned long long value)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void small(double *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 0.5 + 1.0;
}

int main()
{
    const size_t n = (size_t)8256 * 8195 / 2; // one 8192^2 field in d2 elements
    d2 *a, *b;
    double *f, *x;
    unsigned long long *flag, *flag2;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMalloc(&f, 1 << 20));
    CK(hipMalloc(&x, 1 << 20));
    CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void **)&flag2, 8, hipMallocSignalMemory));
    CK(hipMemset(a, 0, n * 16));
    CK(hipMemset(flag, 0, 8));
    CK(hipMemset(flag2, 0, 8));
    CK(hipMemset(f, 0, 1 << 20));
    CK(hipMemset(x, 0, 1 << 20));
    hipStream_t s, side, side_hi;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&side_hi, hipStreamNonBlocking, hi));
    hipEvent_t e1, e2, f1, f2, t0, t1;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    // no system-scope fence when the event is recorded: enough for dependencies inside one device
    CK(hipEventCreateWithFlags(&f1, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&f2, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    const unsigned g = (unsigned)((n + 255) / 256);
    const int iters = 60;
    unsigned long long tick = 0;
    auto run = [&](const char *name, auto body) {
        std::vector<float> ts;
        for (int rep = 0; rep < 5; rep++) {
            for (int k = 0; k < 5; k++) body();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s));
            for (int k = 0; k < iters; k++) body();
            CK(hipEventRecord(t1, s));
            CK(hipEventSynchronize(t1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            ts.push_back(ms / iters * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-86s %8.2f us/iter\n", name, ts[2]);
        fflush(stdout);
        return ts[2];
    };
    auto B = [&] { big<<<g, 256, 0, s>>>(a, b, n); };
    auto F = [&] { small<<<64, 256, 0, s>>>(f, 16384); };
    auto X = [&](hipStream_t q) { small<<<8, 256, 0, q>>>(x, 2048); };
    run("P0  big", [&] { B(); });
    run("P1  frame; big", [&] { F(); B(); });
    run("P2  frame; record(e1); big", [&] { F(); CK(hipEventRecord(e1, s)); B(); });
    run("P2b frame; big; record(e1)", [&] { F(); B(); CK(hipEventRecord(e1, s)); });
    run("P3  frame; record e1; side: wait e1, X, record e2; big; wait e2   (today's step)", [&] {
        F(); CK(hipEventRecord(e1, s)); CK(hipStreamWaitEvent(side, e1, 0)); X(side); CK(hipEventRecord(e2, side));
        B(); CK(hipStreamWaitEvent(s, e2, 0));
    });
    run("P2f frame; record(f1: no system fence); big", [&] { F(); CK(hipEventRecord(f1, s)); B(); });
    run("P2g frame; big; record(f1: no system fence)", [&] { F(); B(); CK(hipEventRecord(f1, s)); });
    run("P3f today's step with no-system-fence events", [&] {
        F(); CK(hipEventRecord(f1, s)); CK(hipStreamWaitEvent(side, f1, 0)); X(side); CK(hipEventRecord(f2, side));
        B(); CK(hipStreamWaitEvent(s, f2, 0));
    });
    run("P3g frame; record f1; side: wait f1, X, record f2; big; wait f2  -- X = 40 us kernel", [&] {
        F(); CK(hipEventRecord(f1, s)); CK(hipStreamWaitEvent(side, f1, 0));
        big<<<(unsigned)(g / 8), 256, 0, side>>>(a + n / 2, b + n / 2, n / 8); CK(hipEventRecord(f2, side));
        B(); CK(hipStreamWaitEvent(s, f2, 0));
    });
    run("P3e the same with ordinary events                                   -- X = 40 us kernel", [&] {
        F(); CK(hipEventRecord(e1, s)); CK(hipStreamWaitEvent(side, e1, 0));
        big<<<(unsigned)(g / 8), 256, 0, side>>>(a + n / 2, b + n / 2, n / 8); CK(hipEventRecord(e2, side));
        B(); CK(hipStreamWaitEvent(s, e2, 0));
    });
    run("P3h the same, high-priority side stream", [&] {
        F(); CK(hipEventRecord(e1, s)); CK(hipStreamWaitEvent(side_hi, e1, 0)); X(side_hi); CK(hipEventRecord(e2, side_hi));
        B(); CK(hipStreamWaitEvent(s, e2, 0));
    });
    run("P4  big; wait e2(previous); frame; record e1; side: wait e1, X, record e2   (deferred join)", [&] {
        B(); CK(hipStreamWaitEvent(s, e2, 0)); F(); CK(hipEventRecord(e1, s)); CK(hipStreamWaitEvent(side, e1, 0));
        X(side); CK(hipEventRecord(e2, side));
    });
    run("P5  big+signal; side: waitValue(flag), X, record e2; main: wait e2   (frame fused, flag edge)", [&] {
        tick++;
        big_signal<<<g, 256, 0, s>>>(a, b, n, flag, tick);
        CK(hipStreamWaitValue64(side, flag, tick, hipStreamWaitValueGte, ~0ull));
        X(side); CK(hipEventRecord(e2, side)); CK(hipStreamWaitEvent(s, e2, 0));
    });
    run("P0  big (again)", [&] { B(); });
    return 0;
}
