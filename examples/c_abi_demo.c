/* The C ABI used from plain C: no Python, no Fortran, no torch -- what any host language's FFI binds.
 * Builds a 1000 x 600 T field the way the reference lays it out (grid_mod.f90:349-385 extents,
 * field_mod.f90:563-624 bounds), runs ten Jacobi steps and a checksum on the device, moves a patch through
 * the two device-sync callbacks, and prints the numbers tests/test_c_abi_demo.py compares with the oracle.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Ldl_esm_inf_amd/lib -ldlesm_hip \
 *       -Wl,-rpath,$PWD/dl_esm_inf_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 */
#include <stdio.h>
#include <stdlib.h>

#include "dlesm_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != DLESM_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dlesm_last_error());       \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 1000, ny = argc > 2 ? atoi(argv[2]) : 600;
    const int nsteps = argc > 3 ? atoi(argv[3]) : 10;

    /* host-side index maps: one tile, depth-1 halo */
    dlesm_decomp decomp;
    dlesm_subdomain sub;
    CHECK(dlesm_decompose(nx, ny, 1, 0, 0, 1, &decomp, &sub));
    int alignment = 1, ld = 0, nyarr = 0;
    CHECK(dlesm_alignment_from_env(&alignment));
    CHECK(dlesm_grid_extents(sub.global.nx, sub.global.ny, alignment, &ld, &nyarr));
    dlesm_region internal, whole;
    CHECK(dlesm_field_bounds(DLESM_T_POINTS, DLESM_OFFSET_NE, DLESM_BC_EXTERNAL, DLESM_BC_EXTERNAL, &sub.internal, ld,
                             nyarr, &internal, &whole));
    printf("G: grid %d %d internal %d %d %d %d\n", ld, nyarr, internal.xstart, internal.xstop, internal.ystart,
           internal.ystop);

    if (dlesm_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "the device entry points have no CPU fallback");
        return 2;
    }
    CHECK(dlesm_init(0));
    dlesm_field *fa = NULL, *fb = NULL;
    CHECK(dlesm_field_create(ld, nyarr, &fa));
    CHECK(dlesm_field_create(ld, nyarr, &fb));
    double *a = dlesm_field_data(fa), *b = dlesm_field_data(fb);
    /* counter-hash initial condition on the whole region (ring included), same ring in both buffers */
    CHECK(dlesm_hash_init_f64(a, ld, nyarr, whole.xstart, whole.xstop, whole.ystart, whole.ystop, 20261004ull, 0, 0, NULL));
    CHECK(dlesm_copy_patch_f64(a, b, ld, nyarr, 1, 1, 1, 1, ld, nyarr, NULL));
    for (int s = 0; s < nsteps; s++) {
        CHECK(dlesm_stencil5_f64(a, b, ld, nyarr, internal.xstart, internal.xstop, internal.ystart, internal.ystop, NULL));
        double *t = a; a = b; b = t;
        dlesm_field *tf = fa; fa = fb; fb = tf;
    }
    double cs = 0.0;
    CHECK(dlesm_checksum_f64(a, ld, nyarr, internal.xstart, internal.xstop, internal.ystart, internal.ystop, &cs, NULL));
    printf("G: checksum %.17e\n", cs);

    /* the reference's device-sync callbacks: read a 3 x 2 patch at (2,2) back into a host array */
    double *host = (double *)calloc((size_t)ld * nyarr, sizeof(double));
    if (!host) return 3;
    dlesm_read_from_device(fa, host, 2, 2, 3, 2, true);
    printf("G: patch %.17e %.17e %.17e %.17e\n", host[(size_t)1 * ld + 1], host[(size_t)1 * ld + 3],
           host[(size_t)2 * ld + 1], host[(size_t)2 * ld + 3]);
    free(host);
    CHECK(dlesm_field_destroy(fa));
    CHECK(dlesm_field_destroy(fb));
    CHECK(dlesm_finalize());
    return 0;
}
