// Which part of a forked capture (caller's stream -> side stream -> back) the HIP runtime of this
// image accepts: plain kernels with the library's event flags, then an RCCL loop-back send/recv.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/graphprobe.hip -o scratch/graphprobe -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define NK(x) do { ncclResult_t e = (x); if (e != ncclSuccess) { printf("%s -> %s (line %d)\n", #x, ncclGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void add1(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;       // bit0: special event flags, bit1: priority side stream, bit2: RCCL, bit3: two fork/joins
    const int n = 1 << 16;
    double *a, *b;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (mode & 2) { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi)); }
    else CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    const unsigned fl = (mode & 1) ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDisableTiming;
    CK(hipEventCreateWithFlags(&e1, fl)); CK(hipEventCreateWithFlags(&e2, fl));
    ncclComm_t comm = nullptr;
    if (mode & 4) {
        ncclUniqueId id; NK(ncclGetUniqueId(&id)); NK(ncclCommInitRank(&comm, 1, id, 0));
        NK(ncclGroupStart()); NK(ncclRecv(b, n, ncclDouble, 0, comm, side)); NK(ncclSend(a, n, ncclDouble, 0, comm, side)); NK(ncclGroupEnd());
        CK(hipStreamSynchronize(side));
    }
    printf("mode %d: begin capture\n", mode); fflush(stdout);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int rep = 0; rep < ((mode & 8) ? 2 : 1); rep++) {       // bit3: the same fork/join twice, events reused
        add1<<<n / 256, 256, 0, s>>>(a, n);
        CK(hipEventRecord(e1, s));
        CK(hipStreamWaitEvent(side, e1, 0));
        if (mode & 4) { NK(ncclGroupStart()); NK(ncclRecv(b, n, ncclDouble, 0, comm, side)); NK(ncclSend(a, n, ncclDouble, 0, comm, side)); NK(ncclGroupEnd()); }
        else add1<<<n / 256, 256, 0, side>>>(b, n);
        CK(hipEventRecord(e2, side));
        add1<<<n / 256, 256, 0, s>>>(a + n / 2, n / 2);
        CK(hipStreamWaitEvent(s, e2, 0));
    }
    printf("mode %d: end capture\n", mode); fflush(stdout);
    CK(hipStreamEndCapture(s, &g));
    printf("mode %d: instantiate\n", mode); fflush(stdout);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 3; k++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double h[2]; CK(hipMemcpy(h, a, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h + 1, b, 8, hipMemcpyDeviceToHost));
    printf("mode %d: ok a[0]=%g b[0]=%g\n", mode, h[0], h[1]);
    return 0;
}
