/*
 * dlesm_hip.h -- C ABI of the MI355X-native dl_esm_inf hot path.
 *
 * Everything a host language binds is declared here: plain C, plain pointers
 * and sizes, no C++ or torch types.  The Fortran side binds these through
 * ISO_C_BINDING (dl_esm_inf_amd/fortran/dlesm_hip_mod.f90), Python through
 * ctypes (dl_esm_inf_amd/_cabi.py).  INTEGRATION.md shows the stub a
 * maintainer of stfc/dl_esm_inf would add.
 *
 * Each entry cites the reference interface it replaces; paths are relative to
 * the reference's finite_difference/src/ directory.
 *
 * Conventions
 *   - all grid indices are 1-based and inclusive, exactly as in the Fortran;
 *   - a field is a column-major array data(1:ld, 1:ny) of doubles
 *     (field_mod.f90:350); element (ji,jj) lives at (jj-1)*ld + (ji-1);
 *   - functions return 0 on success, a negative DLESM_E* code otherwise;
 *     dlesm_last_error() gives the text.  Nothing falls back to the CPU:
 *     device entry points fail with DLESM_ENODEV when there is no GPU.
 *   - `stream` arguments are hipStream_t handles passed as void*; NULL is the
 *     HIP null stream.  Device entry points are asynchronous on that stream
 *     unless stated otherwise.
 *   - hipGraph capture: the kernel entries, the halo exchange and the distributed steps may be
 *     called on a stream that is being captured (hipStreamBeginCapture), so that a whole time loop
 *     replays as one graph launch.  Planning calls, checksums, gathers and anything documented as
 *     synchronising may not.  Under capture a distributed step forks to the library's side stream
 *     and joins back inside the graph (the *_pipelined form then equals the joined form); run the
 *     call once uncaptured first -- pack buffers are allocated on first use.  On a plan whose
 *     mailboxes are connected (peer transport, section 5) the captured operations are the ordinary
 *     single launches -- no RCCL call in the graph, *_pipelined stays the time-loop form -- and a
 *     graph has to hold an EVEN number of mailbox operations of a plan (see there).
 */
#ifndef DLESM_HIP_H
#define DLESM_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLESM_VERSION 310   /* round 3: entries added (the GOcean shallow kernels one by one, time_smooth, fused periodic step, plan description; 310: the peer transport, mailbox mode, the board); no signature changed */

/* error codes */
#define DLESM_OK 0
#define DLESM_EINVAL (-1)   /* bad argument / shape mismatch              */
#define DLESM_ENODEV (-2)   /* no usable HIP device                       */
#define DLESM_EHIP (-3)     /* a HIP runtime call failed                  */
#define DLESM_ERCCL (-4)    /* an RCCL call failed                        */
#define DLESM_EABORT (-5)   /* the reference aborts (gocean_stop) here    */
#define DLESM_ECOMMS (-12)  /* comm list overflow, parallel_comms_mod.f90:1223 */

/* enumerations shared with the Fortran API (values are the reference's) */
enum { DLESM_U_POINTS = 0, DLESM_V_POINTS = 1, DLESM_T_POINTS = 2,
       DLESM_F_POINTS = 3, DLESM_ALL_POINTS = 4 };            /* field_mod.f90:47-52 */
enum { DLESM_OFFSET_SW = 0, DLESM_OFFSET_SE = 1, DLESM_OFFSET_NW = 2,
       DLESM_OFFSET_NE = 3, DLESM_OFFSET_ANY = 4 };           /* grid_mod.f90:52-60  */
enum { DLESM_BC_PERIODIC = 0, DLESM_BC_EXTERNAL = 1, DLESM_BC_NONE = 2 }; /* grid_mod.f90:64-69 */
/* halo-exchange direction codes, parallel_comms_mod.f90:101-110 */
enum { DLESM_IPLUS = 1, DLESM_IMINUS = 2, DLESM_JPLUS = 3, DLESM_JMINUS = 4,
       DLESM_IPLUSJPLUS = 5, DLESM_IMINUSJMINUS = 6, DLESM_IPLUSJMINUS = 7,
       DLESM_IMINUSJPLUS = 8 };

/* region_mod.f90:7-12 -- member order matches the Fortran type so that
 * type(region_type) can be passed by reference. */
typedef struct dlesm_region {
    int nx, ny;
    int xstart, xstop;
    int ystart, ystop;
} dlesm_region;

/* decomposition_mod.f90:44-50 */
typedef struct dlesm_subdomain {
    dlesm_region global;   /* internal part in global coordinates; nx,ny = WHOLE extent */
    dlesm_region internal; /* internal part in local coordinates                        */
} dlesm_subdomain;

/* scalar part of decomposition_mod.f90:54-68 */
typedef struct dlesm_decomp {
    int global_nx, global_ny; /* size of the decomposed domain        */
    int nx, ny;               /* grid of subdomains                   */
    int ndomains;
    int max_width, max_height;
} dlesm_decomp;

#define DLESM_MAXCOMM 16 /* parallel_comms_mod.f90:70 */

/* One rank's message lists: the public tables of parallel_comms_mod.f90:71-83,156-161.
 * destination/source are 0-based ranks, coordinates are local 1-based. */
typedef struct dlesm_comm_tables {
    int nsend, nrecv;
    int dirsend[DLESM_MAXCOMM], destination[DLESM_MAXCOMM];
    int isrcsend[DLESM_MAXCOMM], jsrcsend[DLESM_MAXCOMM];
    int idessend[DLESM_MAXCOMM], jdessend[DLESM_MAXCOMM];
    int nxsend[DLESM_MAXCOMM], nysend[DLESM_MAXCOMM];
    int dirrecv[DLESM_MAXCOMM], source[DLESM_MAXCOMM];
    int isrcrecv[DLESM_MAXCOMM], jsrcrecv[DLESM_MAXCOMM];
    int idesrecv[DLESM_MAXCOMM], jdesrecv[DLESM_MAXCOMM];
    int nxrecv[DLESM_MAXCOMM], nyrecv[DLESM_MAXCOMM];
} dlesm_comm_tables;

/* ------------------------------------------------------------------------
 * 1. Host-side index maps (no GPU needed; bit-exact with the reference)
 * ---------------------------------------------------------------------- */

/* grid_init's DL_ESM_ALIGNMENT block, grid_mod.f90:349-363: parse the
 * environment variable.  *alignment = 1 when unset.  DLESM_EABORT when the
 * reference would gocean_stop (more than 3 characters, not a positive int). */
int dlesm_alignment_from_env(int *alignment);

/* grid_mod.f90:364-385: allocated extents of every field on a subdomain whose
 * whole size is sub_global_nx x sub_global_ny. */
int dlesm_grid_extents(int sub_global_nx, int sub_global_ny, int alignment, int *nx, int *ny);

/* set_field_bounds + c{u,v,t,f}_{sw,ne}_init + field_init, field_mod.f90:563-1122.
 * DLESM_EABORT for the combinations on which the reference stops. */
int dlesm_field_bounds(int grid_points, int offset, int bc_x, int bc_y,
                       const dlesm_region *subdomain_internal, int grid_nx, int grid_ny,
                       dlesm_region *internal, dlesm_region *whole);

/* init_periodic_bc_halos, field_mod.f90:1394-1464: the (source, destination) regions of the
 * periodic-boundary copies of a field with the given internal region, in the reference's order
 * (x pair first, then the y pair over the widened columns).  source/dest hold 4 entries;
 * *num_halos = 0, 2 or 4.  The region's nx, ny members are filled with the extents. */
int dlesm_periodic_halos(const dlesm_region *internal, int bc_x, int bc_y, dlesm_region *source,
                         dlesm_region *dest, int *num_halos);

/* go_decompose, parallel_mod.f90:70-332.  ntilex,ntiley <= 0 selects the
 * reference's automatic tiling.  subdomains must hold ndomains entries. */
int dlesm_decompose(int domainx, int domainy, int ndomains, int ntilex, int ntiley,
                    int halo_width, dlesm_decomp *decomp, dlesm_subdomain *subdomains);

/* iprocmap, parallel_comms_mod.f90:1365-1398 (1-based owner, 0 if none) */
int dlesm_iprocmap(const dlesm_decomp *decomp, const dlesm_subdomain *subdomains,
                   int nranks, int ia, int ja);

/* map_comms, parallel_comms_mod.f90:178-1172, for rank `rank1` (1-based, as
 * get_rank() returns it, parallel_utils_mod.f90:84). */
int dlesm_map_comms(const dlesm_decomp *decomp, const dlesm_subdomain *subdomains,
                    int nranks, int rank1, dlesm_comm_tables *tables);

/* Extension (the reference aborts beyond MAX_HALO_DEPTH = 1, parallel_comms_mod.f90:48,
 * 220-222): tables of a depth-`depth` exchange -- same neighbours, direction codes and order,
 * strips `depth` cells deep against the internal region, corners depth x depth.  Needs a
 * decomposition made with halo_width >= depth and tiles at least `depth` wide and high. */
int dlesm_map_comms_depth(const dlesm_decomp *decomp, const dlesm_subdomain *subdomains,
                          int nranks, int rank1, int depth, dlesm_comm_tables *tables);

/* ------------------------------------------------------------------------
 * 2. Runtime
 * ---------------------------------------------------------------------- */

const char *dlesm_last_error(void);
int dlesm_version(void);

/* number of visible HIP devices (0 when there is none; never an error) */
int dlesm_device_count(void);

/* Bind this process to a device -- the HIP counterpart of
 * acc_init(acc_device_nvidia) in gocean_initialise, gocean_mod.F90:31-33.
 * Creates the library's side stream and events.  Idempotent. */
int dlesm_init(int device);
int dlesm_finalize(void);

/* ------------------------------------------------------------------------
 * 3. Device-resident fields and the reference's device-sync callbacks
 * ---------------------------------------------------------------------- */

/* What r2d_field%device_ptr (field_mod.f90:153) points at: an opaque
 * descriptor that remembers the device buffer and its row stride, which the
 * C-flavour callbacks below are not told. */
typedef struct dlesm_field dlesm_field;

/* allocate ld*ny doubles on the device and zero them (field_mod.f90:350,375) */
int dlesm_field_create(int ld, int ny, dlesm_field **field);
/* describe device memory owned by the caller (e.g. a torch tensor) */
int dlesm_field_wrap(void *device_data, int ld, int ny, dlesm_field **field);
int dlesm_field_destroy(dlesm_field *field);
double *dlesm_field_data(const dlesm_field *field); /* raw device pointer */
int dlesm_field_ld(const dlesm_field *field);
int dlesm_field_ny(const dlesm_field *field);

/* read_from_device_c_interface / write_to_device_c_interface,
 * field_mod.f90:65-73 and 86-94, with exactly that argument list:
 *   read : from = device_ptr (dlesm_field*), to = C_LOC(host data)
 *   write: from = C_LOC(host data),          to = device_ptr (dlesm_field*)
 * startx,starty are 1-based, nx,ny the extent of the patch.  Assign them to
 * fld%read_from_device_c / fld%write_to_device_c.  Errors abort the process
 * (the reference's error model: gocean_stop, gocean_mod.F90:50-57). */
void dlesm_read_from_device(void *from, void *to, int startx, int starty, int nx, int ny,
                            bool blocking);
void dlesm_write_to_device(void *from, void *to, int startx, int starty, int nx, int ny,
                           bool blocking);
/* wait for non-blocking transfers issued by the two callbacks */
int dlesm_transfer_sync(void);

/* ------------------------------------------------------------------------
 * 4. Kernels over a 1-based inclusive index box (the PSy-layer loop nest
 *    `do jj = ystart,ystop ; do ji = xstart,xstop ; call kern_code(ji,jj,...)`,
 *    form: infrastructure_mod.f90:32-41, bounds: field_mod.f90:116-119).
 *    Raw device pointers; ld,ny describe every array passed.
 * ---------------------------------------------------------------------- */

/* out(ji,jj) = 0.25*((in(ji-1,jj)+in(ji+1,jj)) + (in(ji,jj-1)+in(ji,jj+1))) */
int dlesm_stencil5_f64(const double *in, double *out, int ld, int ny,
                       int xstart, int xstop, int ystart, int ystop, void *stream);

/* Optional planning call for dlesm_stencil5_f64 (in the manner of an FFT plan): times about a dozen
 * launch shapes (waves per workgroup, tiles per row, rows per tile) on the caller's own arrays -- each
 * is the same valid step in -> out, the results do not depend on the shape -- and remembers the fastest
 * for later calls with this (ld, box).  Synchronises `stream`.  Without it a fitted rule picks the shape. */
int dlesm_stencil5_autotune_f64(const double *in, double *out, int ld, int ny,
                                int xstart, int xstop, int ystart, int ystop, void *stream);
/* What the planning call kept for this (ld, box): waves per workgroup, wave tiles per row, rows per
 * tile (all 0 when no planning call has been made for it), and whether `out` is stored non-temporally
 * (by size: on from 150 MB per array, when a ping-pong pair no longer fits the 256 MB Infinity Cache).  Host only; for logs and profiles. */
int dlesm_stencil5_planned_shape(int ld, int xstart, int xstop, int ystart, int ystop,
                                 int *waves_per_group, int *tiles_per_row, int *rows_per_tile,
                                 int *nt_stores);

/* TWO Jacobi steps in one sweep (temporal blocking; SURVEY section 8 f.4 -- an extension,
 * the reference stops at MAX_HALO_DEPTH = 1, parallel_comms_mod.f90:48):
 *   t   = J(in) on the intermediate box (exstart:exstop, eystart:eystop), in elsewhere
 *   out = J(t)  on the box (xstart:xstop, ystart:ystop)
 * bit-identical to two dlesm_stencil5_f64 calls through a buffer that equals `in` outside the
 * intermediate box.  One tile: intermediate box = box (fixed boundary ring).  Distributed:
 * the box grown by one cell towards each neighbouring tile, `in` holding depth-2 halos. */
int dlesm_stencil5_x2_f64(const double *in, double *out, int ld, int ny,
                          int xstart, int xstop, int ystart, int ystop,
                          int exstart, int exstop, int eystart, int eystop, void *stream);

/* nsteps (2..8) Jacobi steps in one sweep:
 *   t_0 = in;  t_s = J(t_{s-1}) on the stage box E_s, t_{s-1} elsewhere (s = 1..nsteps-1);
 *   out = J(t_{nsteps-1}) on the box.
 * (exstart:exstop, eystart:eystop) is the LAST stage box E_{nsteps-1}; E_s is that box grown by
 * (nsteps-1-s) cells on every side whose grow_* flag is 1 (W, E, S, N = towards lower i, higher
 * i, lower j, higher j).  One tile: stage box = box, flags 0.  A tile with neighbours: stage
 * box = box grown by 1 towards each neighbour, flag 1 there, `in` holding depth-nsteps halos.
 * Bit-identical to nsteps dlesm_stencil5_f64 calls. */
int dlesm_stencil5_multi_f64(const double *in, double *out, int ld, int ny, int nsteps,
                             int xstart, int xstop, int ystart, int ystop,
                             int exstart, int exstop, int eystart, int eystop,
                             int grow_w, int grow_e, int grow_s, int grow_n, void *stream);

/* A general 9-point (3 x 3) weighted stencil -- the loop nest of any kernel of the form
 *   out(ji,jj) = SUM_{dj,di = -1..1} coef(di,dj) * in(ji+di, jj+dj)
 * (go_arg(GO_READ, GO_CT, GO_STENCIL(111,111,111)) + nine real scalars, argument_mod.f90:39-112).
 * coef[(dj+1)*3 + (di+1)]: south-west, south, south-east, west, centre, east, north-west, north,
 * north-east -- the storage order of a Fortran coef(-1:1,-1:1).  Five-point kernels pass zero corners.
 * Evaluation order: out = ((c_sw*sw + c_s*s) + c_se*se  +  (c_w*w + c_c*c) + c_e*e)
 * + ((c_nw*nw + c_n*n) + c_ne*ne), each row left to right, rows south to north, no FMA contraction. */
int dlesm_stencil9_f64(const double *in, double *out, const double *coef, int ld, int ny,
                       int xstart, int xstop, int ystart, int ystop, void *stream);

/* The same loop nest for a kernel that requests a GRID PROPERTY: the T-point land/sea mask
 * (GO_GRID_MASK_T in the kernel metadata, argument_mod.f90:75-112; the PSy layer passes
 * grid%tmask, here its device mirror grid%tmask_device, grid_mod.f90:104-106).  tmask is a
 * default-integer array with the field layout, > 0 = wet.  Dry points carry their value over;
 * a dry neighbour of a wet point is mirrored (no-flux coast):
 *   out = tmask(ji,jj) > 0 ? 0.25*((w+e)+(s+n)) : in(ji,jj),  w = tmask(ji-1,jj) > 0 ? in(ji-1,jj) : in(ji,jj), ... */
int dlesm_stencil5_masked_f64(const double *in, double *out, const int *tmask, int ld, int ny,
                              int xstart, int xstop, int ystart, int ystop, void *stream);

/* A kernel on all three C-grid point types that also takes a double-precision GRID PROPERTY
 * (GO_GRID_AREA_T, argument_mod.f90:75-112; the PSy layer passes grid%area_t, on the device its mirror
 * grid%area_t_device, grid_mod.f90:104-150): the free-surface (continuity) update of a NEMOLite2D-class
 * model over the box,
 *   r1 = (sshn_u(ji,jj)+hu(ji,jj))*un(ji,jj)         r2 = the same at (ji-1,jj)
 *   r3 = (sshn_v(ji,jj)+hv(ji,jj))*vn(ji,jj)         r4 = the same at (ji,jj-1)
 *   ssha(ji,jj) = sshn_t(ji,jj) + (((r2 - r1) + r4) - r3) * rdt / area_t(ji,jj)
 * (specification frozen in DESIGN.md section 5.10; the reference holds no such loop).  72 B/cell.
 * ssha may alias sshn_t only. */
int dlesm_continuity_f64(double rdt, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         const double *sshn_t, const double *sshn_u, const double *sshn_v,
                         const double *hu, const double *hv, const double *un, const double *vn,
                         const double *area_t, double *ssha, void *stream);

/* Shallow-water u/v/h update: reads u,v,p (3x3 footprint)
 * and uold,vold,pold, writes unew,vnew,pnew on the box. */
typedef struct dlesm_sw_params {
    double fsdx, fsdy;            /* 4/dx, 4/dy            */
    double tdts8, tdtsdx, tdtsdy; /* tdt/8, tdt/dx, tdt/dy */
} dlesm_sw_params;
int dlesm_shallow_step_f64(const dlesm_sw_params *params, int ld, int ny,
                           int xstart, int xstop, int ystart, int ystop,
                           const double *u, const double *v, const double *p,
                           const double *uold, const double *vold, const double *pold,
                           double *unew, double *vnew, double *pnew, void *stream);

/* Optional planning call for dlesm_shallow_step_f64, like dlesm_stencil5_autotune_f64: times a
 * dozen launch shapes and the cache policies of the once-read / once-written arrays on the caller's
 * own arrays (each trial is the same valid step) and remembers the fastest for this (ld, box).
 * Synchronises `stream`. */
int dlesm_shallow_autotune_f64(const dlesm_sw_params *params, int ld, int ny,
                               int xstart, int xstop, int ystart, int ystop,
                               const double *u, const double *v, const double *p,
                               const double *uold, const double *vold, const double *pold,
                               double *unew, double *vnew, double *pnew, void *stream);

/* The SW-offset form of the same step -- the staggering of the GOcean `shallow` benchmark
 * (DESIGN.md section 6.2): u(i,j) on the WEST face of T(i,j), v on the south face, z on the SW
 * corner.  The only staggering the reference supports with periodic boundaries, and only serially
 * (field_mod.f90:675-751, grid_mod.f90:437-442); u, v, p must hold valid periodic halos
 * (dlesm_periodic_halos_apply_f64), the new fields get theirs from the caller the same way. */
int dlesm_shallow_step_sw_f64(const dlesm_sw_params *params, int ld, int ny,
                              int xstart, int xstop, int ystart, int ystop,
                              const double *u, const double *v, const double *p,
                              const double *uold, const double *vold, const double *pold,
                              double *unew, double *vnew, double *pnew, void *stream);

/* planning call for the SW-offset step (dlesm_shallow_step_sw_f64 and its periodic form), as
 * dlesm_shallow_autotune_f64 is for the NE one; the new fields receive one valid step (no periodic copies) */
int dlesm_shallow_autotune_sw_f64(const dlesm_sw_params *params, int ld, int ny,
                                  int xstart, int xstop, int ystart, int ystop,
                                  const double *u, const double *v, const double *p,
                                  const double *uold, const double *vold, const double *pold,
                                  double *unew, double *vnew, double *pnew, void *stream);

/* The same step over the INTERNAL region of periodic fields, with the periodic copies of the new level done by
 * the same launch: every cell stored on the first / last internal column or row is also stored into the halo cell
 * that init_periodic_bc_halos (field_mod.f90:1394-1464) would copy it to, corner halos included.  Leaves unew,
 * vnew, pnew exactly as dlesm_shallow_step_sw_f64 over `internal` followed by
 * dlesm_periodic_halos_apply_multi_f64 does -- one launch instead of three.  bc_x, bc_y: the grid's
 * boundary_conditions(1:2); a non-periodic direction gets no copies. */
int dlesm_shallow_step_sw_periodic_f64(const dlesm_sw_params *params, int ld, int ny,
                                       const dlesm_region *internal, int bc_x, int bc_y,
                                       const double *u, const double *v, const double *p,
                                       const double *uold, const double *vold, const double *pold,
                                       double *unew, double *vnew, double *pnew, void *stream);

/* TWO leapfrog steps per launch (NE offset, fixed boundary ring): level n+1 into unew / vnew / pnew and
 * level n+2 into unew2 / vnew2 / pnew2, bit for bit what
 *     dlesm_shallow_step_f64(..., u, v, p, uold, vold, pold, unew, vnew, pnew);
 *     dlesm_shallow_step_f64(..., unew, vnew, pnew, u, v, p, unew2, vnew2, pnew2);
 * leave behind -- six arrays read and six written per TWO steps, 48 B/cell/step instead of 72.  As for those two calls the ring of
 * unew, vnew, pnew outside the box (which no step writes) must hold the boundary values before the call.  Twelve distinct
 * arrays (level n+2 cannot overwrite level n-1 in place: the first stage reads it one cell around each tile).  The PSy-layer
 * loop nests it replaces: two passes of the un-fused GOcean kernel sequence (infrastructure_mod.f90:13-41 for the form). */
int dlesm_shallow_step_x2_f64(const dlesm_sw_params *q, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                              const double *u, const double *v, const double *p, const double *uold, const double *vold,
                              const double *pold, double *unew, double *vnew, double *pnew, double *unew2, double *vnew2,
                              double *pnew2, void *stream);
/* ... and with the Asselin filter after EACH step: two whole time steps of the GOcean loop (update + time_smooth of the old level,
 * twice) in one launch.  In: level n (u, v, p) and the filtered level n-1 (uold, vold, pold), both untouched.  Out: level n+2 in
 * unew2 / vnew2 / pnew2 and the FILTERED level n+1 in uold2 / vold2 / pold2 -- the new current and old levels two calls of
 * dlesm_shallow_step_smooth_f64 with the usual rotation leave, bit for bit, provided the boundary ring is the same at every time
 * level (the ring of the unfiltered level n+1, which no array holds, is taken from u, v, p).  48 B/cell/step against 96 for the
 * one-launch filtered step.  Time loop: ping-pong (u.., uold..) <-> (unew2.., uold2..). */
int dlesm_shallow_step_smooth_x2_f64(const dlesm_sw_params *q, double alpha, int ld, int ny, int xstart, int xstop, int ystart,
                                     int ystop, const double *u, const double *v, const double *p, const double *uold,
                                     const double *vold, const double *pold, double *unew2, double *vnew2, double *pnew2,
                                     double *uold2, double *vold2, double *pold2, void *stream);
/* The same two entries for the SW-offset, doubly periodic model -- the configuration of the GOcean `shallow` benchmark
 * (field_mod.f90:675-751, 1394-1464): == two calls of dlesm_shallow_step_sw_periodic_f64 / dlesm_shallow_step_sw_smooth_periodic_f64
 * with the loop's rotation, periodic images of every level that comes out included.  Level n+1 one cell outside the box is the
 * image of level n+1 inside, so the first stage reads level n TWO cells outside the box from where it is the image of.  Beyond
 * the single steps' preconditions: level n-1 carries valid periodic halos as well. */
int dlesm_shallow_step_sw_x2_periodic_f64(const dlesm_sw_params *q, int ld, int ny, const dlesm_region *internal, int bc_x, int bc_y,
                                          const double *u, const double *v, const double *p, const double *uold, const double *vold,
                                          const double *pold, double *unew, double *vnew, double *pnew, double *unew2,
                                          double *vnew2, double *pnew2, void *stream);
int dlesm_shallow_step_sw_smooth_x2_periodic_f64(const dlesm_sw_params *q, double alpha, int ld, int ny, const dlesm_region *internal,
                                                 int bc_x, int bc_y, const double *u, const double *v, const double *p,
                                                 const double *uold, const double *vold, const double *pold, double *unew2,
                                                 double *vnew2, double *pnew2, double *uold2, double *vold2, double *pold2,
                                                 void *stream);

/* One WHOLE time step of the GOcean leapfrog in one launch: the u/v/h update AND the Asselin filter of the old level
 * (the benchmark's time_smooth kernel), from values the lanes already hold --
 *     unew, vnew, pnew <- step(u, v, p, uold, vold, pold);   uold <- u + alpha*(unew - 2*u + uold)   (likewise vold, pold; in place)
 * -- bit for bit what dlesm_shallow_step_f64 followed by three dlesm_time_smooth_f64 calls leave, at 96 B/cell
 * (six arrays read, six written) instead of 72 + 3 x 32 = 168.  After the call the host program rotates
 * u <- unew (uold already holds the filtered u; the former u buffers are free).  The nine arrays must be distinct. */
int dlesm_shallow_step_smooth_f64(const dlesm_sw_params *params, double alpha, int ld, int ny,
                                  int xstart, int xstop, int ystart, int ystop,
                                  const double *u, const double *v, const double *p,
                                  double *uold, double *vold, double *pold,
                                  double *unew, double *vnew, double *pnew, void *stream);
/* ... and the SW-offset form over the internal region of periodic fields, the periodic images of the new level AND of
 * the filtered old level written by the same launch: a whole time step of the GOcean `shallow` benchmark = ONE launch. */
int dlesm_shallow_step_sw_smooth_periodic_f64(const dlesm_sw_params *params, double alpha, int ld, int ny,
                                              const dlesm_region *internal, int bc_x, int bc_y,
                                              const double *u, const double *v, const double *p,
                                              double *uold, double *vold, double *pold,
                                              double *unew, double *vnew, double *pnew, void *stream);

/* The GOcean `shallow` kernel set as SEPARATE launch entries: one per PSy loop nest, which is what a
 * PSyclone-generated PSy layer has -- `do jj / do ji / call compute_cu_code(ji, jj, cu%data, p%data,
 * u%data)` becomes dlesm_compute_cu_f64 over the same index box (kernel form infrastructure_mod.f90:13-41,
 * metadata argument_mod.f90:39-112, kernel_mod.f90:28-50).  Array arguments in the kernels' own order
 * (the written field first).  `offset` is the kernel's index_offset, DLESM_OFFSET_NE or DLESM_OFFSET_SW;
 * formulas frozen in DESIGN.md section 6 (NE), 6.2 (SW), 6.3 (time_smooth) -- the expression trees of the
 * fused step, so that the seven launches
 *     cu, cv, z, h  (each over the box grown towards its consumers, or over the internal region followed by
 *     the periodic copies)  then  unew, vnew, pnew
 * give bit for bit what dlesm_shallow_step_f64 / dlesm_shallow_step_sw_f64 give, at 224 B/cell of HBM
 * traffic instead of 72.  The box plus the cells its stencil reads must lie inside the ld x ny arrays;
 * the output may not alias an input (time_smooth updates field_old in place).
 *   NE:  cu(i,j) = 0.5*(p(i+1,j)+p(i,j))*u(i,j)         SW:  cu(i,j) = 0.5*(p(i,j)+p(i-1,j))*u(i,j)   ... */
int dlesm_compute_cu_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         double *cu, const double *p, const double *u, void *stream);
int dlesm_compute_cv_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                         double *cv, const double *p, const double *v, void *stream);
int dlesm_compute_z_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                        double fsdx, double fsdy, double *z, const double *p, const double *u,
                        const double *v, void *stream);
int dlesm_compute_h_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                        double *h, const double *p, const double *u, const double *v, void *stream);
int dlesm_compute_unew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                           double tdts8, double tdtsdx, double *unew, const double *uold,
                           const double *z, const double *cv, const double *h, void *stream);
int dlesm_compute_vnew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                           double tdts8, double tdtsdy, double *vnew, const double *vold,
                           const double *z, const double *cu, const double *h, void *stream);
int dlesm_compute_pnew_f64(int offset, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                           double tdtsdx, double tdtsdy, double *pnew, const double *pold,
                           const double *cu, const double *cv, void *stream);
/* time_smooth (the Asselin filter of the `shallow` leapfrog; any offset; pointwise):
 *   field_old(i,j) = field(i,j) + alpha*(field_new(i,j) - 2.0*field(i,j) + field_old(i,j)) */
int dlesm_time_smooth_f64(int ld, int ny, int xstart, int xstop, int ystart, int ystop, double alpha,
                          const double *field, const double *field_new, double *field_old, void *stream);

/* All periodic-boundary copies of one field (dlesm_periodic_halos' regions, applied in order with
 * the patch copy below), enqueued on `stream`: what the PSy layer of a periodic model does after
 * every kernel that writes the field. */
int dlesm_periodic_halos_apply_f64(double *field, int ld, int ny, const dlesm_region *internal,
                                   int bc_x, int bc_y, void *stream);
/* the same for up to 16 fields of one shape and internal region in two launches (all x copies, then
 * all y copies): what follows a step that writes unew, vnew and pnew */
int dlesm_periodic_halos_apply_multi_f64(double *const *fields, int nfields, int ld, int ny,
                                         const dlesm_region *internal, int bc_x, int bc_y, void *stream);

/* field_copy_code over a box (infrastructure_mod.f90:32-41) and the patch copy
 * used for periodic boundaries (copy_2dfield_patch, field_mod.f90:1179-1187):
 * dst(dx0.., dy0..) = src(sx0.., sy0..) for an nx x ny patch. */
int dlesm_copy_patch_f64(const double *src, double *dst, int ld, int ny_arr,
                         int sx0, int sy0, int dx0, int dy0, int nx, int ny, void *stream);
/* set_field, field_mod.f90:1191-1202, restricted to a box */
int dlesm_fill_f64(double *f, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                   double value, void *stream);
/* array_checksum's local part, SUM(ABS(f(xs:xe,ys:ye))), field_mod.f90:1298-1302.
 * Synchronous: returns the value in *result (host). Deterministic tree order. */
int dlesm_checksum_f64(const double *f, int ld, int ny, int xstart, int xstop,
                       int ystart, int ystop, double *result, void *stream);
/* The same sum with NO host synchronisation: *result_dev -- device memory, or host memory the device can write
 * (hipHostMalloc) -- receives the value when `stream` gets there; bit-identical to dlesm_checksum_f64.  For time
 * loops that record a checksum every few steps (field_checksum of field_mod.f90:1209-1219 costs the reference a
 * full device-to-host copy of the field per call, :538); combine across ranks afterwards (dlesm_global_sum_f64). */
int dlesm_checksum_async_f64(const double *f, int ld, int ny, int xstart, int xstop,
                             int ystart, int ystop, double *result_dev, void *stream);
/* synthetic initial condition of BASELINE.md: f(i,j) = u01(splitmix64(seed ^ (gi + gj<<32)))
 * on the box, gi = gx0+i-1, gj = gy0+j-1; cells outside the box are left alone. */
int dlesm_hash_init_f64(double *f, int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                        uint64_t seed, int64_t gx0, int64_t gy0, void *stream);

/* (The measured copy ceilings of bench.py live in their own library, include/dlesm_lab.h: measurement tooling.)
 *
 * Settings, by name; returns the previous value (0 if the key had none).  Three classes (dlesm_tuning_class):
 *   0  USER  what a host program may want to say ("Settings"): dm_safe, dm_wait_seconds, dm_peer,
 *            dm_peer_exchange, mailbox_fences, mailbox_fields, mailbox_gather_host, dm_acquire, j5_dm_corners,
 *            j5_nt_stores, j5_use_tuned, side_stream_priority, dm_graph_force.
 *   1  HOOK  forces a path the library takes BY ITSELF for some inputs (an unaligned base, a capture, a planned launch
 *            shape, DLESM_DM_SAFE ...) so that tests can reach it on any input; every setting computes the same bits.
 *   2  LAB   selects a comparison-only kernel or a diagnostic that skips work: compiled into libdlesm_hip_lab.so only
 *            (-DDLESM_LAB, the same sources); libdlesm_hip.so ignores such a key and says so once on stderr.
 * An unknown key is reported on stderr (once) and kept. */
int dlesm_set_tuning(const char *key, int value);
/* 0 USER, 1 HOOK, 2 LAB, -1 unknown */
int dlesm_tuning_class(const char *key);
/* 1 in libdlesm_hip_lab.so (built with -DDLESM_LAB), 0 in the product library */
int dlesm_is_lab_build(void);

/* ------------------------------------------------------------------------
 * 5. Device-resident halo exchange over RCCL
 *    (replaces r2d_field%halo_exchange -> exchange_generic,
 *     field_mod.f90:1231-1256, parallel_comms_mod.f90:1501-1855, and the
 *     MPI calls of parallel/parallel_utils_mod.f90:148-183)
 * ---------------------------------------------------------------------- */

#define DLESM_UNIQUE_ID_BYTES 128
/* rank 0 creates the id, the host program distributes it (file, MPI, torch store ...) */
int dlesm_comm_unique_id(void *id /* DLESM_UNIQUE_ID_BYTES */);
/* parallel_init, parallel_utils_mod.f90:77-90: rank0 is 0-based here */
int dlesm_comm_init(const void *id, int nranks, int rank0);
/* MAILBOX MODE -- the same job without a communication library: no RCCL communicator is created.  Every message plan
 * connects its mailboxes when it is created (dlesm_halo_plan_create becomes collective; room for mailbox_fields /
 * DLESM_MAILBOX_FIELDS fields, default 3), halo exchanges and the distributed Jacobi / shallow-water steps go through them
 * (stores over xGMI, see the peer transport below), dlesm_global_sum_f64 is eight bytes per rank over the host-side board
 * (summed in rank order, the same bits on every rank), dlesm_gather_f64 / dlesm_gather_inner_f64 copy every rank's
 * block straight into a gather buffer the library owns on the root (exported once, mapped once per job by each rank).  `id`: a session name all ranks share, made by rank 0 with
 * dlesm_board_nonce and handed round like an RCCL id (dlesm_rendezvous_publish / _fetch, a torch store ...).
 * Replaces MPI_Init + the MPI calls of parallel_utils_mod.f90:77-255 for a job of one process per GPU on one node. */
int dlesm_comm_init_mailbox(const void *id, int nranks, int rank0);
int dlesm_comm_is_mailbox(void);
/* How often, in this process, hipIpcOpenMemHandle had to be tried again (or a gather fell back to host memory).  Processes
 * of one node import in turns (flock on /dev/shm/dlesm-ipc-<uid>.lock), so anything but 0 is a finding: every occurrence is
 * also logged to stderr with the HIP error name.  IPC mappings need HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every
 * process on this driver ("Environment"). */
int dlesm_ipc_open_retries(void);
int dlesm_comm_finalize(void);
int dlesm_comm_rank(void);   /* 0-based, -1 before init */
int dlesm_comm_size(void);

/* File rendezvous for the id (dl_esm_inf_amd/csrc/dlesm_rendezvous.cpp): what a job started by
 * a launcher that only exports RANK/WORLD_SIZE uses instead of MPI_Init
 * (parallel/parallel_utils_mod.f90:77-90).  Host only, no GPU needed.  Rank 0: remove(path),
 * dlesm_comm_unique_id, publish(path, id, token) -- atomic (rename).  Others: fetch waits up to
 * timeout_ms for a record whose job token matches and whose publisher started within
 * DLESM_RENDEZVOUS_SLACK_S (default 120) seconds of the caller; anything else is a stale file
 * of a dead job and is ignored.  token: at most 111 characters. */
int dlesm_rendezvous_remove(const char *path);
int dlesm_rendezvous_publish(const char *path, const void *id, const char *token);
int dlesm_rendezvous_fetch(const char *path, void *id, const char *token, int timeout_ms);
/* dry runs only (no communicator, hence no ncclCommInitRank to hold rank 0 until everybody has the
 * id): readers acknowledge, rank 0 waits for nranks-1 acknowledgements and removes them */
int dlesm_rendezvous_ack(const char *path, int rank0);
int dlesm_rendezvous_wait_acks(const char *path, int nranks, int timeout_ms);

/* The BOARD: a host-side all-gather between the processes of one job through files (/dev/shm, or DLESM_BOARD_DIR) -- the
 * control plane of mailbox mode.  Host only.  nonce: a session name (letters, digits, '-', '_') no other job has;
 * allgather: every rank contributes `bytes`, `all` receives nranks x bytes in rank order (may be null on ranks that only
 * contribute); every rank must make the same sequence of calls.  A rank that does not show up within
 * DLESM_BOARD_TIMEOUT_S (default 600) makes the call fail. */
int dlesm_board_nonce(void *id /* DLESM_UNIQUE_ID_BYTES */);
int dlesm_board_open(const void *id, int nranks, int rank0);
int dlesm_board_is_open(void);
int dlesm_board_allgather(const void *mine, size_t bytes, void *all);
int dlesm_board_close(void);
/* parallel_abort in mailbox mode: leaves a note; ranks waiting on the board for this rank fail with its text (DLESM_EABORT)
 * instead of sitting out the time-out.  The note stays (it names a dead job's session, which no other job shares). */
int dlesm_board_abort(const char *msg);

/* Message plan for fields of shape (ld, ny): device copy of the tables, pack
 * buffers for the strided (east/west) strips.  One plan serves every field of
 * that shape (all dl_esm_inf fields share the grid's extents, field_mod.f90:327-333). */
typedef struct dlesm_halo_plan dlesm_halo_plan;
int dlesm_halo_plan_create(const dlesm_comm_tables *tables, int ld, int ny,
                           dlesm_halo_plan **plan);
int dlesm_halo_plan_destroy(dlesm_halo_plan *plan);

/* One ncclRecv / ncclSend of an exchange, as dlesm_halo_plan_describe reports it. */
typedef struct dlesm_msg_desc {
    int is_recv;           /* 1 = ncclRecv, 0 = ncclSend                                                   */
    int peer;              /* 0-based rank                                                                 */
    int dir;               /* direction code of the strip (DLESM_IPLUS ...)                                */
    int field;             /* which of the nfields fields; -1 = the strips of ALL fields, field after field */
    int i0, j0, nx, ny;    /* the strip inside the field, 1-based origin                                   */
    long count;            /* doubles on the wire                                                          */
    long buffer_offset;    /* doubles from the start of the staging buffer (receive buffer for a receive,
                              send buffer for a send); -1 = the message travels in place                  */
} dlesm_msg_desc;

/* The calls ONE exchange of `nfields` fields under `dirs_mask` makes on a plan created from `tables` for fields of
 * ld x ny, in ISSUE ORDER -- the receives then the sends, each sorted by (peer, direction code); aggregated = 1: the
 * one-message-per-neighbour-and-direction form (dlesm_halo_exchange_multi_f64, dlesm_halo_exchange_f64 on its own,
 * the shallow-water step); 0: the Jacobi step's own form (rows in place, strided strips through the pack buffer;
 * field-major).  RCCL has no tags (the reference matches messages by tag_orig + dir, parallel_comms_mod.f90:1606,
 * 1647): between a pair of ranks the k-th send meets the k-th receive, so this order IS the protocol.  Host only (no
 * GPU, no communicator); built and walked by the very code the plan and its exchanges use.  *n_out = number of
 * calls; with max_out = 0 only the count is returned. */
int dlesm_halo_plan_describe(const dlesm_comm_tables *tables, int ld, int ny, int nfields,
                             unsigned dirs_mask, int aggregated, dlesm_msg_desc *out, int max_out,
                             int *n_out);

/* halo_exchange(depth=1) of one device field: pack -> grouped ncclSend/ncclRecv
 * -> unpack, all enqueued on `stream`.  dirs_mask selects the enabled edge
 * directions (bit d-1 for DLESM_IPLUS..DLESM_JMINUS: the comm1..comm4 arguments of
 * exchange_generic); diagonals are enabled when both their edges are,
 * parallel_comms_mod.f90:1557-1571.  DLESM_DIRS_ALL is what halo_exchange passes
 * (field_mod.f90:1247-1248); 0 exchanges nothing, as in the reference; halos of a
 * disabled direction are left untouched.  DLESM_DIRS_NO_DIAGONALS (an extension)
 * keeps the four corner messages off whatever the edges say: all a 5-point stencil needs.
 * One exchange per plan may be in flight at a time (the pack buffers are the plan's). */
#define DLESM_DIRS_ALL 0xFu
#define DLESM_DIRS_NO_DIAGONALS 0x10u
int dlesm_halo_exchange_f64(dlesm_halo_plan *plan, double *field, unsigned dirs_mask,
                            void *stream);

/* The same for several fields of the plan's shape at once: ONE grouped launch for all of them
 * (what a time step that updates u, v and p needs), ONE message per neighbour and direction that
 * carries the strips of all the fields, field after field (an RCCL send/recv pair costs microseconds
 * whatever its size: message count is the lever).  At most 16 fields. */
int dlesm_halo_exchange_multi_f64(dlesm_halo_plan *plan, double *const *fields, int nfields,
                                  unsigned dirs_mask, void *stream);

/* One distributed Jacobi time step with the exchange hidden behind the
 * interior: frame(out) on `stream` (its west/east columns written straight into the
 * send buffer too), then [exchange(out) on the library's side stream] ||
 * [interior(out) on `stream`], joined on `stream`.  On return (asynchronously)
 * `out` holds the new values AND valid depth-1 EDGE halos (west, east, south, north:
 * DLESM_DIRS_ALL | DLESM_DIRS_NO_DIAGONALS -- the 5-point stencil never reads a corner
 * halo cell, so the corner messages are not sent), i.e. it is ready to be the `in` of
 * the next step.  `in` must have valid edge halos. */
int dlesm_jacobi5_step_dm(dlesm_halo_plan *plan, const double *in, double *out,
                          int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                          void *stream);

/* The same step for a TIME LOOP of such steps: it returns with the exchange of `out` still in flight
 * on the library's side stream -- the caller's stream carries one kernel launch per step and no
 * event wait.  The next pipelined step on the same plan and stream waits for that exchange on the
 * device (its frame workgroups, the only ones that read halos, sleep on a flag the side stream sets
 * once the messages have landed; bounded; the received west/east strips are read straight from the
 * receive buffer, not unpacked into the field); every other entry point that takes the plan joins it
 * first -- the join also runs the deferred unpack, so that `out` then holds valid edge halos (the
 * outputs of EARLIER steps of the loop keep whatever west/east halos they had: only the joined
 * output's are promised).  Before
 * anything ELSE reads the halo cells of `out` (a kernel of the host program, a copy on another
 * stream), call dlesm_halo_plan_join.  Same results as dlesm_jacobi5_step_dm, bit for bit. */
int dlesm_jacobi5_step_dm_pipelined(dlesm_halo_plan *plan, const double *in, double *out,
                                    int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                                    void *stream);
/* order `stream` behind the exchange a pipelined step left in flight (no-op when there is none) */
int dlesm_halo_plan_join(dlesm_halo_plan *plan, void *stream);

/* ---- Peer transport of the distributed Jacobi step: the stencil kernel itself is the exchange. ----------------------
 * What the reference does with MPI_Isend / MPI_Irecv / MPI_Waitany per strip (parallel_comms_mod.f90:1601-1750,
 * parallel_utils_mod.f90:148-211) and the default path here with one RCCL group per step, a connected plan does with
 * stores: the frame workgroups of the step launch write every cell a neighbour needs straight into that neighbour's
 * receive MAILBOX (peer-mapped fine-grained memory: xGMI stores) and then raise the neighbour's arrival flag; they read
 * their own halo operands from the local mailbox once its arrival flags are up.  No RCCL kernel, no side stream, no
 * pack, no event.  Mailboxes are double-buffered on the step number, which therefore has to advance on every rank alike
 * (each rank takes the same sequence of distributed steps on its plan -- what a halo exchange asks for anyway).
 * Results are bit for bit those of the RCCL path.  dlesm_jacobi5_step_dm / _pipelined use the mailboxes (nfields >= 1), and
 * so do dlesm_shallow_step_dm / _pipelined / _smooth_dm* when the plan was connected with nfields >= 3 (plans of halo depth 1,
 * stepped over the internal region), and dlesm_halo_exchange_f64 / _multi_f64 themselves (any depth, nfields <= the
 * count the plan was connected for: one launch that copies the send strips into the neighbours' mailboxes and raises their
 * flags, one that waits for this rank's flags and unpacks; dm_peer_exchange = 0 keeps the RCCL group for them).  The
 * fused multi-step and 3x3 distributed steps keep their RCCL-side machinery.  The mailboxes of a plan are one resource:
 * operations on it issued on different streams are ordered one behind the other by the library (an event); ordering the
 * FIELDS between streams stays the caller's business, as with any other entry.  dm_peer = 0 (dlesm_set_tuning)
 * switches a connected plan back to RCCL -- on every rank or on none.
 * hipGraph capture: the number of a mailbox operation is kept in DEVICE memory (each operation's flag-raising workgroup
 * writes the next operation's number), the host only tracks its parity -- which mailbox half the operation's pointers
 * address.  Mailbox operations may therefore be captured, and replayed any number of times between un-captured ones,
 * provided (a) a graph holds an EVEN number of them per plan and ends joined (dlesm_halo_plan_join inside the capture
 * after *_pipelined steps), (b) every replay starts at the parity the capture started at (an even number of un-captured
 * operations in between), (c) the operation before the capture was issued on the capturing stream, and (d) every rank
 * replays alike.  A replay out of step is detected on the device (last number raised != this one - 1) and raises the
 * process-wide flag of dlesm_wait_timed_out: wrong halos cannot leave silently.
 *
 * Connecting is collective over the ranks that share neighbours:
 *   1. dlesm_halo_plan_peer_export(plan, my_rank, nfields = 1, blob): allocates this rank's mailbox and writes
 *      DLESM_PEER_BLOB_BYTES describing it (an IPC handle + the slot of each receive message);
 *   2. the host program all-gathers the blobs in rank order (MPI_Allgather, torch.distributed.all_gather, a file ...);
 *   3. dlesm_halo_plan_peer_connect(plan, my_rank, nranks, blobs): maps the neighbours' mailboxes
 *      (hipIpcOpenMemHandle; a rank that is its own neighbour uses the pointer) and matches every send with the receive
 *      it meets -- the k-th message to a neighbour is the k-th that neighbour receives from this rank.
 * dlesm_halo_plan_peer_connect_rccl does 1-3 with ncclAllGather on the library's communicator.
 * my_rank / the peers in the plan's tables are 0-based ranks, as everywhere in this header. */
#define DLESM_PEER_BLOB_BYTES 1024
int dlesm_halo_plan_peer_export(dlesm_halo_plan *plan, int my_rank, int nfields, void *blob);
int dlesm_halo_plan_peer_connect(dlesm_halo_plan *plan, int my_rank, int nranks, const void *blobs);
int dlesm_halo_plan_peer_connect_rccl(dlesm_halo_plan *plan, int nfields);
int dlesm_halo_plan_peer_connected(const dlesm_halo_plan *plan);   /* 1 / 0 */
/* HOST ONLY (no device, no communicator), for checking the matching on any mesh: the blob a plan made from `tables` would
 * export (without an IPC handle), and, given the blobs of all ranks, what every SEND of this rank is matched with -- the code
 * dlesm_halo_plan_peer_export / _connect run.  One record per send, in the plan's (peer, direction) order. */
typedef struct dlesm_peer_match_desc {
    int peer, dir;         /* the neighbour (0-based rank) and the direction code of the send                   */
    int i0, j0, nx, ny;    /* the strip it reads in this rank's field (1-based origin, extent)                 */
    long count;            /* cells per field                                                                  */
    int slot;              /* index of the matching receive in the NEIGHBOUR's (peer, direction)-sorted list   */
    long off;              /* its per-field offset in the neighbour's mailbox parity                           */
} dlesm_peer_match_desc;
int dlesm_peer_blob_describe(const dlesm_comm_tables *tables, int ld, int ny, int my_rank, int nfields, void *blob);
int dlesm_peer_match_describe(const dlesm_comm_tables *tables, int ld, int ny, int my_rank, int nranks, int nfields,
                              const void *blobs, dlesm_peer_match_desc *out, int max_out, int *n_out);

/* Device-side waits of the distributed steps are bounded: 30 s for this GPU's own frame workgroups, and
 * dm_wait_seconds (dlesm_set_tuning; default 600, 0 = no limit, as the reference waits in MPI_Waitany,
 * parallel_comms_mod.f90:1773-1798) wherever the wait is for an EXCHANGE, i.e. for the slowest neighbour.  A wait
 * that gives up raises ONE process-wide flag; the stream then runs on, so whatever was enqueued behind the wait may
 * have read halos that never arrived.  From that moment EVERY device entry point of this library (steps, checksum,
 * gather, field callbacks ...) fails with DLESM_EHIP: wrong numbers cannot leave silently.  Returns 1 if the flag is
 * up; clear != 0 acknowledges it (after the host program has destroyed its halo plans) and makes the library
 * re-measure whether streams still run side by side. */
int dlesm_wait_timed_out(int clear);

/* The one-launch and time-loop forms park a waiting kernel on the library's side stream while the kernel that
 * releases it runs on the caller's stream: that needs kernels of the two streams to execute side by side (not so
 * under tools that serialise kernels).  Measured once per caller's stream -- at dlesm_halo_plan_create for the null
 * stream, else at the first step on a stream: one device synchronisation, 64 bytes, at most 50 ms -- and remembered.
 * This call measures again NOW (after re-creating a stream, attaching a tool ...): 1 = side by side (one-launch
 * forms are used), 0 = not (the steps take their event-ordered form), < 0 on error.  Not inside a graph capture. */
int dlesm_probe_stream_concurrency(void *stream);

/* The distributed step of any 3 x 3 weighted kernel (dlesm_stencil9_f64's coefficients and
 * evaluation order): out = stencil9(in) on the box, then the halos of `out` valid as after
 * out%halo_exchange(1) -- all eight directions when a corner weight is non-zero (corner halos are
 * operands of the next step), the four edges when all four are zero (the diagonal halo cells are
 * then left untouched).  `in` must hold valid halos for the same directions.  The frame of `out` is
 * computed first (west/east columns straight into the send buffer), the exchange runs on the
 * library's side stream beside the interior sweep, the call returns with `stream` ordered behind
 * both.  Bit-identical to dlesm_stencil9_f64 + dlesm_halo_exchange_f64. */
int dlesm_stencil9_step_dm(dlesm_halo_plan *plan, const double *in, double *out, const double *coef,
                           int ld, int ny, int xstart, int xstop, int ystart, int ystop, void *stream);

/* nsteps (2..8) distributed Jacobi time steps per call, ONE depth-nsteps exchange per call
 * (temporal blocking across tiles; dlesm_stencil5_multi_f64 with stage boxes grown towards
 * every neighbouring tile).  `plan` must come from dlesm_map_comms_depth(depth = nsteps) tables
 * and `in` must hold valid depth-nsteps halos; `out` leaves with valid depth-nsteps halos.
 * Bit-identical to nsteps x (dlesm_stencil5_f64 + depth-nsteps exchange). */
int dlesm_jacobi5_multi_step_dm(dlesm_halo_plan *plan, const double *in, double *out,
                                int ld, int ny, int nsteps,
                                int xstart, int xstop, int ystart, int ystop, void *stream);

/* The distributed form of dlesm_shallow_step_f64: frame of unew/vnew/pnew, then one grouped
 * exchange of the three new fields on the side stream behind the interior.  u, v, p must have
 * valid halos; unew, vnew, pnew leave with valid halos (corners included). */
int dlesm_shallow_step_dm(dlesm_halo_plan *plan, const dlesm_sw_params *params, int ld, int ny,
                          int xstart, int xstop, int ystart, int ystop,
                          const double *u, const double *v, const double *p,
                          const double *uold, const double *vold, const double *pold,
                          double *unew, double *vnew, double *pnew, void *stream);

/* The same step for a TIME LOOP (the contract of dlesm_jacobi5_step_dm_pipelined): returns with the
 * exchange of unew/vnew/pnew in flight on the side stream; the next pipelined step on the same plan
 * and stream -- which reads them as u/v/p -- waits for it on the device, in its frame workgroups;
 * every other entry that takes the plan joins first; dlesm_halo_plan_join orders a stream behind it.
 * Between pipelined steps the caller only rotates the nine pointers.  Same results, bit for bit. */
int dlesm_shallow_step_dm_pipelined(dlesm_halo_plan *plan, const dlesm_sw_params *params, int ld, int ny,
                                    int xstart, int xstop, int ystart, int ystop,
                                    const double *u, const double *v, const double *p,
                                    const double *uold, const double *vold, const double *pold,
                                    double *unew, double *vnew, double *pnew, void *stream);

/* The two distributed forms of dlesm_shallow_step_smooth_f64 (the step with the Asselin filter of the old level folded
 * in): dlesm_shallow_step_dm[_pipelined]'s frame / exchange / interior, uold, vold, pold filtered in place.  The filtered
 * old level needs no exchange: the next step reads it at (i, j) only.  Same results as dlesm_shallow_step_dm[_pipelined]
 * followed by three dlesm_time_smooth_f64 calls over the box, bit for bit. */
int dlesm_shallow_step_smooth_dm(dlesm_halo_plan *plan, const dlesm_sw_params *params, double alpha, int ld, int ny,
                                 int xstart, int xstop, int ystart, int ystop,
                                 const double *u, const double *v, const double *p,
                                 double *uold, double *vold, double *pold,
                                 double *unew, double *vnew, double *pnew, void *stream);
int dlesm_shallow_step_smooth_dm_pipelined(dlesm_halo_plan *plan, const dlesm_sw_params *params, double alpha,
                                           int ld, int ny, int xstart, int xstop, int ystart, int ystop,
                                           const double *u, const double *v, const double *p,
                                           double *uold, double *vold, double *pold,
                                           double *unew, double *vnew, double *pnew, void *stream);

/* global_sum, parallel_utils_mod.f90:230-238: in-place sum of one host double
 * over all ranks (synchronous). */
int dlesm_global_sum_f64(double *value);
/* gather, parallel_utils_mod.f90:242-255: n doubles per rank (device memory)
 * -> n*nranks doubles on rank 0 (device memory). Synchronous. */
int dlesm_gather_f64(const double *send, double *recv, int n);

/* ------------------------------------------------------------------------
 * 6. Device-side gather / scatter of whole fields
 *    (gather_inner_data, field_mod.f90:1313-1390; the init_global_data scatter of the
 *     constructor, field_mod.f90:378-389)
 * ---------------------------------------------------------------------- */

/* Pack loop of gather_inner_data (field_mod.f90:1360-1368): the box of a device field, j outer /
 * i inner, into send[0 .. nx*ny); the rest of the slot (tiles are uneven, every rank sends the
 * size of the largest one, field_mod.f90:1348-1351) is zeroed.  slot >= nx*ny. */
int dlesm_pack_inner_f64(const double *field, int ld, int ny, int xstart, int xstop, int ystart,
                         int ystop, double *send, long slot, void *stream);

/* Unpack loop of gather_inner_data on rank 0 (field_mod.f90:1372-1386): slot r of `recv` (slot
 * doubles each) goes to rank r's box in the global array, `global` being a device array of
 * global_nx x global_ny doubles (column-major, like the Fortran result).  One launch for all ranks. */
int dlesm_unpack_gathered_f64(const double *recv, long slot, const dlesm_decomp *decomp,
                              const dlesm_subdomain *subdomains, int nranks, double *global,
                              void *stream);

/* gather_inner_data for a device-resident field: pack on the device, ncclSend/ncclRecv to rank 0
 * (MPI_Gather of parallel_utils_mod.f90:242-255), unpack on the device, ONE device-to-host copy of
 * the global array into global_host (global_nx x global_ny doubles; only read on rank 0, may be
 * NULL elsewhere).  internal: this rank's internal region (halo_x = xstart-1, halo_y = ystart-1,
 * field_mod.f90:1348-1349).  Synchronous.  With one rank: the copy-out of field_mod.f90:1332-1343. */
int dlesm_gather_inner_f64(const double *field, int ld, int ny, const dlesm_region *internal,
                           const dlesm_decomp *decomp, const dlesm_subdomain *subdomains,
                           int nranks, double *global_host);

/* The constructor's scatter (field_mod.f90:378-389): this rank's internal region of the device
 * field <- the matching patch of a HOST global array (global_nx x global_ny), one strided
 * host-to-device copy.  sub: this rank's subdomain (its global box gives the patch). Synchronous. */
int dlesm_scatter_inner_f64(const double *global_host, int global_nx, int global_ny,
                            const dlesm_subdomain *sub, double *field, int ld, int ny);

#ifdef __cplusplus
}
#endif
#endif /* DLESM_HIP_H */
