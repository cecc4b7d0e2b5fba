/* A multi-rank Jacobi job from plain C with NO communication library: one process per rank (they may share a GPU),
 * mailbox mode of libdlesm_hip.so.  What the reference gets from MPI_Init, MPI_Isend/Irecv/Waitany, MPI_Allreduce
 * (parallel_utils_mod.f90:77-255) comes from: a session name handed round through the library's file rendezvous, message
 * plans that connect their mailboxes when they are made, a distributed step whose frame workgroups store into the
 * neighbours' memory, and a global sum over the host-side board.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/mailbox_demo.c \
 *       -Ldl_esm_inf_amd/lib -ldlesm_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/dl_esm_inf_amd/lib -Wl,-rpath,/opt/rocm/lib -o mailbox_demo
 *   for r in 0 1 2 3; do RANK=$r WORLD_SIZE=4 MASTER_PORT=29700 ./mailbox_demo 600 600 24 & done; wait
 *
 * NX x NY is the GLOBAL domain; the decomposition is go_decompose's.  Prints "G: checksum <initial> <final>" on rank 0:
 * the same numbers (to the rounding of the rank-order sum) whatever the number of ranks.
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
            fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, rc_, dlesm_last_error()); \
            dlesm_board_abort(dlesm_last_error());                                   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static int env_int(const char *name, int fallback)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : fallback;
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 600, ny = argc > 2 ? atoi(argv[2]) : 600;
    const int nsteps = argc > 3 ? atoi(argv[3]) : 24;
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
    if (world < 1 || world > 64 || rank < 0 || rank >= world) {
        fprintf(stderr, "RANK / WORLD_SIZE\n");
        return 1;
    }
    if (dlesm_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "the device entry points have no CPU fallback");
        return 2;
    }
    CHECK(dlesm_init(env_int("LOCAL_RANK", rank) % dlesm_device_count()));

    /* the session name: made by rank 0, handed round like an RCCL id */
    if (world > 1) {
        char path[256], token[100];
        unsigned char id[DLESM_UNIQUE_ID_BYTES];
        snprintf(path, sizeof path, "/dev/shm/dlesm_mailbox_demo_%d", env_int("MASTER_PORT", 29700));
        const char *job = getenv("DLESM_JOB_ID");                /* a launcher's run id keeps a dead job's record out */
        snprintf(token, sizeof token, "%d:mailbox-demo:%.40s", world, job ? job : "-");
        if (rank == 0) {
            CHECK(dlesm_rendezvous_remove(path));
            CHECK(dlesm_board_nonce(id));
            CHECK(dlesm_rendezvous_publish(path, id, token));
        } else {
            CHECK(dlesm_rendezvous_fetch(path, id, token, 120000));
        }
        CHECK(dlesm_comm_init_mailbox(id, world, rank));
        unsigned char every[64];                                /* everybody has the name: rank 0 may remove the file */
        CHECK(dlesm_board_allgather(&id[0], 1, every));
        if (rank == 0) CHECK(dlesm_rendezvous_remove(path));
    }

    /* decomposition, extents, bounds, message tables: the host index maps of the reference */
    dlesm_decomp decomp;
    dlesm_subdomain subs[64];
    CHECK(dlesm_decompose(nx, ny, world, 0, 0, 1, &decomp, subs));
    const dlesm_subdomain sub = subs[rank];
    int alignment = 1, ld = 0, nyarr = 0;
    CHECK(dlesm_alignment_from_env(&alignment));
    CHECK(dlesm_grid_extents(sub.global.nx, sub.global.ny, alignment, &ld, &nyarr));
    dlesm_region it, whole;
    CHECK(dlesm_field_bounds(DLESM_T_POINTS, DLESM_OFFSET_NE, DLESM_BC_EXTERNAL, DLESM_BC_EXTERNAL, &sub.internal, ld, nyarr,
                             &it, &whole));
    dlesm_comm_tables tables;
    memset(&tables, 0, sizeof tables);
    if (world > 1) CHECK(dlesm_map_comms(&decomp, subs, world, rank + 1, &tables));
    dlesm_halo_plan *plan = NULL;
    CHECK(dlesm_halo_plan_create(&tables, ld, nyarr, &plan));    /* collective in mailbox mode: connects the mailboxes */

    dlesm_field *fa = NULL, *fb = NULL;
    CHECK(dlesm_field_create(ld, nyarr, &fa));
    CHECK(dlesm_field_create(ld, nyarr, &fb));
    double *a = dlesm_field_data(fa), *b = dlesm_field_data(fb);
    /* the counter hash of the GLOBAL cell index, the fixed boundary ring included: every decomposition the same field */
    const long gx0 = sub.global.xstart - sub.internal.xstart + 1, gy0 = sub.global.ystart - sub.internal.ystart + 1;
    CHECK(dlesm_hash_init_f64(a, ld, nyarr, it.xstart - 1, it.xstop + 1, it.ystart - 1, it.ystop + 1, 20261004ull, gx0, gy0, NULL));
    CHECK(dlesm_copy_patch_f64(a, b, ld, nyarr, 1, 1, 1, 1, ld, nyarr, NULL));
    double cs0 = 0.0, cs1 = 0.0;
    CHECK(dlesm_checksum_f64(a, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, &cs0, NULL));
    CHECK(dlesm_global_sum_f64(&cs0));

    for (int k = 0; k < nsteps; k++) {                           /* the time-loop form: one launch per step, no join */
        CHECK(dlesm_jacobi5_step_dm_pipelined(plan, a, b, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, NULL));
        double *t = a;
        a = b;
        b = t;
    }
    CHECK(dlesm_halo_plan_join(plan, NULL));
    CHECK(dlesm_checksum_f64(a, ld, nyarr, it.xstart, it.xstop, it.ystart, it.ystop, &cs1, NULL));
    CHECK(dlesm_global_sum_f64(&cs1));
    if (dlesm_wait_timed_out(0)) {
        fprintf(stderr, "rank %d: a device-side wait gave up\n", rank);
        return 3;
    }
    if (rank == 0) printf("G: checksum %.17e %.17e\nG: ranks %d tiles %dx%d mailbox %d\n", cs0, cs1, world, decomp.nx, decomp.ny,
                          dlesm_comm_is_mailbox());
    CHECK(dlesm_halo_plan_destroy(plan));
    CHECK(dlesm_field_destroy(fa));
    CHECK(dlesm_field_destroy(fb));
    CHECK(dlesm_comm_finalize());
    CHECK(dlesm_finalize());
    return 0;
}
