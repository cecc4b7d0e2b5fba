/* A distributed time loop as ONE hipGraph, from plain C: two ping-pong steps of dlesm_jacobi5_step_dm
 * (frame -> [RCCL exchange on the library's side stream || interior sweep] -> join) are captured once
 * and replayed; the result must equal the same steps issued one by one.  One rank, which is its own
 * eight neighbours (a periodic wrap), so that the RCCL group is real.  Needs RCCL >= 2.27.7 (ROCm 7.2):
 * the library refuses the capture on older ones, where hipStreamEndCapture crashes (scripts/graphprobe.hip).
 * A fourth argument `peer` connects the plan's mailboxes first: the captured steps are then single launches
 * whose frame workgroups store into the neighbours' mailboxes -- no RCCL call in the graph, any RCCL version; the sequence
 * numbers of the mailboxes live on the device and advance from replay to replay (a graph holds an even number of steps).
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/graph_demo.c \
 *       -Ldl_esm_inf_amd/lib -ldlesm_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/dl_esm_inf_amd/lib -Wl,-rpath,/opt/rocm/lib -o graph_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "dlesm_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != DLESM_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dlesm_last_error());       \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define HIP(call)                                                                    \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));             \
            return 1;                                                                \
        }                                                                            \
    } while (0)

/* rank 0 as its own eight neighbours: dir, source corner, destination corner, extent */
static void loopback(dlesm_comm_tables *t, const dlesm_region *it)
{
    const int xl = it->xstart, xh = it->xstop, yl = it->ystart, yh = it->ystop;
    const int m[8][7] = {
        {DLESM_IMINUS, xh, yl, xl - 1, yl, 1, it->ny},      {DLESM_IPLUS, xl, yl, xh + 1, yl, 1, it->ny},
        {DLESM_JMINUS, xl, yh, xl, yl - 1, it->nx, 1},      {DLESM_JPLUS, xl, yl, xl, yh + 1, it->nx, 1},
        {DLESM_IMINUSJMINUS, xh, yh, xl - 1, yl - 1, 1, 1}, {DLESM_IPLUSJPLUS, xl, yl, xh + 1, yh + 1, 1, 1},
        {DLESM_IPLUSJMINUS, xl, yh, xh + 1, yl - 1, 1, 1},  {DLESM_IMINUSJPLUS, xh, yl, xl - 1, yh + 1, 1, 1}};
    memset(t, 0, sizeof *t);
    t->nsend = t->nrecv = 8;
    for (int k = 0; k < 8; k++) {
        t->dirsend[k] = t->dirrecv[k] = m[k][0];
        t->isrcsend[k] = t->isrcrecv[k] = m[k][1];
        t->jsrcsend[k] = t->jsrcrecv[k] = m[k][2];
        t->idessend[k] = t->idesrecv[k] = m[k][3];
        t->jdessend[k] = t->jdesrecv[k] = m[k][4];
        t->nxsend[k] = t->nxrecv[k] = m[k][5];
        t->nysend[k] = t->nyrecv[k] = m[k][6];
    }
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 500, ny = argc > 2 ? atoi(argv[2]) : 300;
    const int nsteps = argc > 3 ? atoi(argv[3]) & ~1 : 8;           /* an even number: the graph holds two */
    const int peer = argc > 4 && !strcmp(argv[4], "peer");
    dlesm_decomp decomp;
    dlesm_subdomain sub;
    CHECK(dlesm_decompose(nx, ny, 1, 0, 0, 1, &decomp, &sub));
    int alignment = 1, ld = 0, nyarr = 0;
    CHECK(dlesm_alignment_from_env(&alignment));
    CHECK(dlesm_grid_extents(sub.global.nx, sub.global.ny, alignment, &ld, &nyarr));
    dlesm_region it, whole;
    CHECK(dlesm_field_bounds(DLESM_T_POINTS, DLESM_OFFSET_NE, DLESM_BC_EXTERNAL, DLESM_BC_EXTERNAL, &sub.internal, ld, nyarr,
                             &it, &whole));
    printf("G: grid %d %d\n", ld, nyarr);
    if (dlesm_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "the device entry points have no CPU fallback");
        return 2;
    }
    CHECK(dlesm_init(0));
    unsigned char id[DLESM_UNIQUE_ID_BYTES];
    CHECK(dlesm_comm_unique_id(id));
    CHECK(dlesm_comm_init(id, 1, 0));
    dlesm_comm_tables tables;
    loopback(&tables, &it);
    dlesm_halo_plan *plan = NULL;
    CHECK(dlesm_halo_plan_create(&tables, ld, nyarr, &plan));
    if (peer) CHECK(dlesm_halo_plan_peer_connect_rccl(plan, 1));    /* mailboxes for one field; blobs all-gathered by the library */
    printf("G: transport %s\n", dlesm_halo_plan_peer_connected(plan) ? "mailboxes" : "rccl");
    dlesm_field *fa = NULL, *fb = NULL;
    CHECK(dlesm_field_create(ld, nyarr, &fa));
    CHECK(dlesm_field_create(ld, nyarr, &fb));
    double *a = dlesm_field_data(fa), *b = dlesm_field_data(fb);
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    double sums[2] = {0.0, 0.0};
    float ms[2] = {0.0f, 0.0f};                                     /* device time of the loop, both forms */
    hipEvent_t t0, t1;
    HIP(hipEventCreate(&t0));
    HIP(hipEventCreate(&t1));
    for (int pass = 0; pass < 2; pass++) {                          /* 0: step by step, 1: captured and replayed */
        CHECK(dlesm_hash_init_f64(a, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, 20261004ull, 0, 0, s));
        CHECK(dlesm_halo_exchange_f64(plan, a, DLESM_DIRS_ALL, s));
        CHECK(dlesm_copy_patch_f64(a, b, ld, nyarr, 1, 1, 1, 1, ld, nyarr, s));
        if (pass == 0) {
            HIP(hipEventRecord(t0, s));
            for (int k = 0; k < nsteps; k += 2) {
                CHECK(dlesm_jacobi5_step_dm(plan, a, b, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, s));
                CHECK(dlesm_jacobi5_step_dm(plan, b, a, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, s));
            }
            HIP(hipEventRecord(t1, s));
        } else {
            hipGraph_t graph;
            hipGraphExec_t exec;
            HIP(hipStreamSynchronize(s));
            HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = dlesm_jacobi5_step_dm(plan, a, b, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, s);
            if (rc == DLESM_OK) rc = dlesm_jacobi5_step_dm(plan, b, a, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, s);
            if (rc != DLESM_OK) {                                   /* e.g. an RCCL that cannot be captured */
                fprintf(stderr, "capture refused: %s\n", dlesm_last_error());
                (void)hipStreamEndCapture(s, &graph);
                return 4;
            }
            HIP(hipStreamEndCapture(s, &graph));
            HIP(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
            HIP(hipEventRecord(t0, s));
            for (int k = 0; k < nsteps; k += 2) HIP(hipGraphLaunch(exec, s));
            HIP(hipEventRecord(t1, s));
            HIP(hipStreamSynchronize(s));
            HIP(hipGraphExecDestroy(exec));
            HIP(hipGraphDestroy(graph));
        }
        /* halos included: the exchange inside the steps is part of what is compared */
        CHECK(dlesm_checksum_f64(a, ld, nyarr, whole.xstart, whole.xstop, whole.ystart, whole.ystop, &sums[pass], s));
        HIP(hipEventElapsedTime(&ms[pass], t0, t1));
    }
    printf("G: ms_per_step stepwise %.4f graph %.4f\n", ms[0] / nsteps, ms[1] / nsteps);
    printf("G: stepwise %.17e\nG: graph %.17e\n", sums[0], sums[1]);
    HIP(hipStreamDestroy(s));
    CHECK(dlesm_halo_plan_destroy(plan));
    CHECK(dlesm_field_destroy(fa));
    CHECK(dlesm_field_destroy(fb));
    CHECK(dlesm_comm_finalize());
    CHECK(dlesm_finalize());
    return 0;
}
