/* TEST INFRASTRUCTURE ONLY -- not part of the shipped product.
 *
 * CPU restatement ("oracle") of the dl_esm_inf hot path: the integer maps the
 * reference computes (array extents, field bounds, domain decomposition,
 * halo-exchange message tables), the halo exchange itself, checksum /
 * scatter / gather, and the stencil loops in GOcean kernel form.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/finite_difference/src unless stated).  The restatement is
 * pinned by tests/test_oracle_vs_reference.py against tests/golden/ref_*.json,
 * which were produced by running the real reference (oracle/make_golden.py).
 * Pinning status per function is listed in DESIGN.md section 3:
 *   - grid extents, field bounds, go_decompose, scatter/gather, checksum:
 *       pinned bit-exactly by outputs of the reference run here;
 *   - map_comms / exchange: pinned by the reference's own known-answer tests
 *       (tests/dist_mem/test_halos.f90 hill() property, test_gsum, test_reduction)
 *       and by the message tables recorded in SURVEY.md section 2.1 [probe];
 *   - jacobi5 / shallow-water arithmetic: PARITY UNPINNED -- the reference
 *       contains no stencil loop (they live in PSyclone-generated code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library.
 */
#ifndef DLESM_ORACLE_H
#define DLESM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* region_mod.f90:7-12 */
typedef struct {
    int nx, ny, xstart, xstop, ystart, ystop;
} orc_region;

/* decomposition_mod.f90:44-50 */
typedef struct {
    orc_region global;   /* position of the internal part in the global domain; nx/ny = WHOLE extent */
    orc_region internal; /* internal region in local indices */
} orc_subdomain;

/* decomposition_mod.f90:54-68 (subdomains[] is caller-allocated, ndomains long) */
typedef struct {
    int global_nx, global_ny, nx, ny, ndomains, max_width, max_height;
} orc_decomp;

#define ORC_MAXCOMM 16 /* parallel_comms_mod.f90:70 */

/* parallel_comms_mod.f90:71-83 : one rank's send and receive lists */
typedef struct {
    int nsend, nrecv;
    int dirsend[ORC_MAXCOMM], destination[ORC_MAXCOMM];
    int isrcsend[ORC_MAXCOMM], jsrcsend[ORC_MAXCOMM];
    int idessend[ORC_MAXCOMM], jdessend[ORC_MAXCOMM];
    int nxsend[ORC_MAXCOMM], nysend[ORC_MAXCOMM];
    int dirrecv[ORC_MAXCOMM], source[ORC_MAXCOMM];
    int isrcrecv[ORC_MAXCOMM], jsrcrecv[ORC_MAXCOMM];
    int idesrecv[ORC_MAXCOMM], jdesrecv[ORC_MAXCOMM];
    int nxrecv[ORC_MAXCOMM], nyrecv[ORC_MAXCOMM];
} orc_comms;

/* grid_mod.f90:349-385.  alignment<=0 means DL_ESM_ALIGNMENT unset (=> 1). */
void orc_grid_extents(int sub_global_nx, int sub_global_ny, int alignment, int *nx, int *ny);

/* field_mod.f90:563-1122.  Returns 0, or 1 when the reference aborts (gocean_stop)
 * for this (ptype, offset, bc) combination.  internal/whole get all six members. */
int orc_field_bounds(int ptype, int offset, int bcx, int bcy,
                     const orc_region *sub_internal, int grid_nx, int grid_ny,
                     orc_region *internal, orc_region *whole);

/* parallel_mod.f90:70-332 (auto tiling when ntilex<=0). subs has room for ndom entries. */
void orc_decompose(int domainx, int domainy, int ndom, int ntilex, int ntiley, int hwidth,
                   orc_decomp *d, orc_subdomain *subs);

/* parallel_comms_mod.f90:1365-1398 ; returns 1-based owner or 0 */
int orc_iprocmap(const orc_decomp *d, const orc_subdomain *subs, int nranks, int ia, int ja);

/* parallel_comms_mod.f90:178-1172 ; irank is 1-based. Returns ierr. */
int orc_map_comms(const orc_decomp *d, const orc_subdomain *subs, int nranks, int irank,
                  orc_comms *c);

/* parallel_comms_mod.f90:1501-1855 done for ALL ranks in one process:
 * fields[r] is rank r's array (ld[r] leading dimension), comms[r] its tables.
 * Sends are matched to receives by (source, destination, direction tag),
 * exactly what the MPI tags tag_orig+dir do (pcomms:1606,1647). Returns the
 * number of unmatched messages (0 on success). */
int orc_exchange_all(int nranks, double **fields, const int *ld, const orc_comms *comms);
/* the same with the reference's comm1..comm4 arguments (direction codes 1..4, 0 = unused);
 * no_diagonals != 0 additionally switches the four corner messages off */
int orc_exchange_dirs(int nranks, double **fields, const int *ld, const orc_comms *comms,
                      int comm1, int comm2, int comm3, int comm4, int no_diagonals);

/* field_mod.f90:1298-1302 (local part; SUM(ABS()) evaluated in j-outer,i-inner order) */
double orc_checksum(const double *f, int ld, int xstart, int xstop, int ystart, int ystop);

/* field_mod.f90:378-389 : global -> local scatter into the subdomain's internal region */
void orc_scatter(const double *global, int gnx, const orc_subdomain *sub, double *local, int ld);

/* field_mod.f90:1313-1390 : every rank's internal region -> global array on root */
void orc_gather_all(int nranks, double **fields, const int *ld, const orc_decomp *d,
                    const orc_subdomain *subs, double *global);

/* Deterministic counter-based initial condition:
 * u01(splitmix64(seed ^ (gi + gj*2^32))) in [0,1). gi,gj are global 1-based. */
double orc_hash_u01(uint64_t seed, int64_t gi, int64_t gj);

/* 5-point Jacobi in GOcean kernel form (calling convention of
 * infrastructure_mod.f90:32-41): out(ji,jj) = 0.25*((w+e)+(s+n)) for every
 * (ji,jj) of the 1-based inclusive box. PARITY UNPINNED by the reference. */
void orc_jacobi5(const double *in, double *out, int ld,
                 int xstart, int xstop, int ystart, int ystop);
/* same loops with an OpenMP `parallel for` over jj (what PSyclone's OMP transformation emits) */
/* general 3x3 weighted stencil, coef[(dj+1)*3 + (di+1)] */
void orc_stencil9(const double *in, double *out, const double *coef, int ld,
                  int xstart, int xstop, int ystart, int ystop);
/* continuity (free-surface) update: T, U, V fields + the grid property area_t */
void orc_continuity(double rdt, int ld, int xstart, int xstop, int ystart, int ystop, const double *sshn_t,
                    const double *sshn_u, const double *sshn_v, const double *hu, const double *hv, const double *un,
                    const double *vn, const double *area_t, double *ssha);
/* masked 5-point Jacobi (kernel with a GO_GRID_MASK_T argument); grid_init's tmask fill */
void orc_jacobi5_masked(const double *in, double *out, const int *tmask, int ld,
                        int xstart, int xstop, int ystart, int ystop);
void orc_tmask_fill(const int *user, int user_ld, int nx, int ny, int xstart, int xstop, int ystart,
                    int ystop, int *tmask);
void orc_jacobi5_omp(const double *in, double *out, int ld,
                     int xstart, int xstop, int ystart, int ystop, int nthreads);

/* Shallow-water (Sadourny / GOcean "shallow" form, NE-offset indexing frozen in
 * DESIGN.md section 6).  One time step = the kernels below in order; each is a
 * PSy-style double loop over the 1-based inclusive box. PARITY UNPINNED. */
typedef struct {
    double fsdx, fsdy;            /* 4/dx, 4/dy            */
    double tdts8, tdtsdx, tdtsdy; /* tdt/8, tdt/dx, tdt/dy */
} orc_sw_params;
/* cu,cv,z,h are full-size scratch fields (same ld); unew,vnew,pnew are written on
 * the box only. Intermediates are evaluated on the box grown by one cell where a
 * consumer needs them (all operands stay inside the depth-1 boundary ring). */
void orc_sw_step(const orc_sw_params *p, int ld, int xstart, int xstop, int ystart, int ystop,
                 const double *u, const double *v, const double *pf,
                 const double *uold, const double *vold, const double *pold,
                 double *cu, double *cv, double *z, double *h,
                 double *unew, double *vnew, double *pnew);

void orc_copy_rows_omp(double *dst, const double *src, int ld, int ny, int nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
/* SW-offset shallow-water step and the periodic-boundary halo copies
 * (field_mod.f90:1394-1464, 1179-1187) */
void orc_sw_step_sw(const orc_sw_params *q, int ld, int xs, int xe, int ys, int ye,
                    const double *u, const double *v, const double *p,
                    const double *uold, const double *vold, const double *pold,
                    double *cu, double *cv, double *z, double *h,
                    double *unew, double *vnew, double *pnew);
/* one kernel of the GOcean shallow set as its own PSy loop nest (see dlesm_oracle.c); PARITY UNPINNED */
int orc_sw_kernel(int kernel, int sw_offset, int ld, int xs, int xe, int ys, int ye, double s0, double s1,
                  double *out, const double *a, const double *b, const double *c, const double *d);
int orc_periodic_halos(const orc_region *it, int bc_x, int bc_y, orc_region *source, orc_region *dest);
void orc_apply_periodic_halos(double *f, int ld, const orc_region *it, int bc_x, int bc_y);

#endif
