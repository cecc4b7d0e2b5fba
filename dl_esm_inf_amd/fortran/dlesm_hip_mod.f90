!> ISO_C_BINDING view of include/dlesm_hip.h: the thin layer between the Fortran
!! API modules and libdlesm_hip.so.  One interface block per C entry point, the
!! interoperable mirrors of the C structs, and nothing else.
module dlesm_hip_mod
  use iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: DLESM_OK = 0, DLESM_EINVAL = -1, DLESM_ENODEV = -2, &
       DLESM_EHIP = -3, DLESM_ERCCL = -4, DLESM_EABORT = -5, DLESM_ECOMMS = -12
  integer, parameter :: DLESM_MAXCOMM = 16
  integer, parameter :: DLESM_UNIQUE_ID_BYTES = 128
  !> dirs_mask values of dlesm_halo_exchange_f64 (include/dlesm_hip.h)
  integer(c_int), parameter :: DLESM_DIRS_ALL = 15_c_int, DLESM_DIRS_NO_DIAGONALS = 16_c_int

  !> struct dlesm_region
  type, bind(C) :: c_region
     integer(c_int) :: nx, ny, xstart, xstop, ystart, ystop
  end type c_region

  !> struct dlesm_subdomain
  type, bind(C) :: c_subdomain
     type(c_region) :: global, internal
  end type c_subdomain

  !> struct dlesm_decomp
  type, bind(C) :: c_decomp
     integer(c_int) :: global_nx, global_ny, nx, ny, ndomains, max_width, max_height
  end type c_decomp

  !> struct dlesm_comm_tables
  type, bind(C) :: c_comm_tables
     integer(c_int) :: nsend, nrecv
     integer(c_int), dimension(DLESM_MAXCOMM) :: dirsend, destination, isrcsend, jsrcsend, &
          idessend, jdessend, nxsend, nysend, dirrecv, source, isrcrecv, jsrcrecv, &
          idesrecv, jdesrecv, nxrecv, nyrecv
  end type c_comm_tables

  !> struct dlesm_sw_params
  type, bind(C) :: c_sw_params
     real(c_double) :: fsdx, fsdy, tdts8, tdtsdx, tdtsdy
  end type c_sw_params

  interface
     ! ---- host-side index maps -------------------------------------------
     function dlesm_alignment_from_env(alignment) bind(C, name="dlesm_alignment_from_env") result(rc)
       import :: c_int
       integer(c_int), intent(out) :: alignment
       integer(c_int) :: rc
     end function
     function dlesm_grid_extents(sub_nx, sub_ny, alignment, nx, ny) &
          bind(C, name="dlesm_grid_extents") result(rc)
       import :: c_int
       integer(c_int), value :: sub_nx, sub_ny, alignment
       integer(c_int), intent(out) :: nx, ny
       integer(c_int) :: rc
     end function
     function dlesm_field_bounds(grid_points, offset, bc_x, bc_y, sub_internal, grid_nx, &
          grid_ny, internal, whole) bind(C, name="dlesm_field_bounds") result(rc)
       import :: c_int, c_region
       integer(c_int), value :: grid_points, offset, bc_x, bc_y, grid_nx, grid_ny
       type(c_region), intent(in) :: sub_internal
       type(c_region), intent(out) :: internal, whole
       integer(c_int) :: rc
     end function
     function dlesm_decompose(domainx, domainy, ndomains, ntilex, ntiley, halo_width, &
          decomp, subdomains) bind(C, name="dlesm_decompose") result(rc)
       import :: c_int, c_decomp, c_subdomain
       integer(c_int), value :: domainx, domainy, ndomains, ntilex, ntiley, halo_width
       type(c_decomp), intent(out) :: decomp
       type(c_subdomain), intent(out) :: subdomains(*)
       integer(c_int) :: rc
     end function
     function dlesm_iprocmap(decomp, subdomains, nranks, ia, ja) &
          bind(C, name="dlesm_iprocmap") result(owner)
       import :: c_int, c_decomp, c_subdomain
       type(c_decomp), intent(in) :: decomp
       type(c_subdomain), intent(in) :: subdomains(*)
       integer(c_int), value :: nranks, ia, ja
       integer(c_int) :: owner
     end function
     function dlesm_map_comms_depth(decomp, subdomains, nranks, rank1, depth, tables) &
          bind(C, name="dlesm_map_comms_depth") result(rc)
       import :: c_int, c_decomp, c_subdomain, c_comm_tables
       type(c_decomp), intent(in) :: decomp
       type(c_subdomain), intent(in) :: subdomains(*)
       integer(c_int), value :: nranks, rank1, depth
       type(c_comm_tables), intent(out) :: tables
       integer(c_int) :: rc
     end function
     function dlesm_map_comms(decomp, subdomains, nranks, rank1, tables) &
          bind(C, name="dlesm_map_comms") result(rc)
       import :: c_int, c_decomp, c_subdomain, c_comm_tables
       type(c_decomp), intent(in) :: decomp
       type(c_subdomain), intent(in) :: subdomains(*)
       integer(c_int), value :: nranks, rank1
       type(c_comm_tables), intent(out) :: tables
       integer(c_int) :: rc
     end function
     ! ---- runtime ----------------------------------------------------------
     function dlesm_last_error() bind(C, name="dlesm_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function
     function dlesm_device_count() bind(C, name="dlesm_device_count") result(n)
       import :: c_int
       integer(c_int) :: n
     end function
     function dlesm_init(device) bind(C, name="dlesm_init") result(rc)
       import :: c_int
       integer(c_int), value :: device
       integer(c_int) :: rc
     end function
     function dlesm_finalize() bind(C, name="dlesm_finalize") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function
     ! ---- device fields ----------------------------------------------------
     function dlesm_field_create(ld, ny, field) bind(C, name="dlesm_field_create") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: ld, ny
       type(c_ptr), intent(out) :: field
       integer(c_int) :: rc
     end function
     function dlesm_field_destroy(field) bind(C, name="dlesm_field_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: field
       integer(c_int) :: rc
     end function
     function dlesm_field_data(field) bind(C, name="dlesm_field_data") result(p)
       import :: c_ptr
       type(c_ptr), value :: field
       type(c_ptr) :: p
     end function
     !> the two C-flavour callbacks of field_mod (reference field_mod.f90:65-73,86-94)
     subroutine dlesm_read_from_device(from, to, startx, starty, nx, ny, blocking) &
          bind(C, name="dlesm_read_from_device")
       import :: c_ptr, c_int, c_bool
       type(c_ptr), intent(in), value :: from, to
       integer(c_int), intent(in), value :: startx, starty, nx, ny
       logical(c_bool), intent(in), value :: blocking
     end subroutine
     subroutine dlesm_write_to_device(from, to, startx, starty, nx, ny, blocking) &
          bind(C, name="dlesm_write_to_device")
       import :: c_ptr, c_int, c_bool
       type(c_ptr), intent(in), value :: from, to
       integer(c_int), intent(in), value :: startx, starty, nx, ny
       logical(c_bool), intent(in), value :: blocking
     end subroutine
     function dlesm_transfer_sync() bind(C, name="dlesm_transfer_sync") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function
     ! ---- kernels ------------------------------------------------------------
     function dlesm_stencil5_f64(in, out, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_stencil5_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: in, out, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_stencil9_f64(in, out, coef, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_stencil9_f64") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: in, out, stream
       real(c_double), intent(in) :: coef(9)
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_stencil9_step_dm(plan, in, out, coef, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_stencil9_step_dm") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: plan, in, out, stream
       real(c_double), intent(in) :: coef(9)
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_continuity_f64(rdt, ld, ny, xstart, xstop, ystart, ystop, sshn_t, sshn_u, sshn_v, hu, hv, &
          un, vn, area_t, ssha, stream) bind(C, name="dlesm_continuity_f64") result(rc)
       import :: c_int, c_ptr, c_double
       real(c_double), value :: rdt
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: sshn_t, sshn_u, sshn_v, hu, hv, un, vn, area_t, ssha, stream
       integer(c_int) :: rc
     end function
     function dlesm_stencil5_masked_f64(in, out, tmask, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_stencil5_masked_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: in, out, tmask, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_f64(params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_sw_f64(params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_sw_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_sw_periodic_f64(params, ld, ny, internal, bc_x, bc_y, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_sw_periodic_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_region
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_x2_f64(params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, uold, vold, pold, &
          unew, vnew, pnew, unew2, vnew2, pnew2, stream) bind(C, name="dlesm_shallow_step_x2_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_sw_x2_periodic_f64(params, ld, ny, internal, bc_x, bc_y, u, v, p, uold, vold, pold, &
          unew, vnew, pnew, unew2, vnew2, pnew2, stream) bind(C, name="dlesm_shallow_step_sw_x2_periodic_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_region
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_sw_smooth_x2_periodic_f64(params, alpha, ld, ny, internal, bc_x, bc_y, u, v, p, uold, vold, pold, &
          unew2, vnew2, pnew2, uold2, vold2, pold2, stream) bind(C, name="dlesm_shallow_step_sw_smooth_x2_periodic_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_region, c_double
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_smooth_x2_f64(params, alpha, ld, ny, xstart, xstop, ystart, ystop, u, v, p, uold, vold, pold, &
          unew2, vnew2, pnew2, uold2, vold2, pold2, stream) bind(C, name="dlesm_shallow_step_smooth_x2_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_double
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_smooth_f64(params, alpha, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_smooth_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_double
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_sw_smooth_periodic_f64(params, alpha, ld, ny, internal, bc_x, bc_y, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_sw_smooth_periodic_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_region, c_double
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_smooth_dm(plan, params, alpha, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_smooth_dm") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_double
       type(c_ptr), value :: plan
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_smooth_dm_pipelined(plan, params, alpha, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_smooth_dm_pipelined") result(rc)
       import :: c_int, c_ptr, c_sw_params, c_double
       type(c_ptr), value :: plan
       type(c_sw_params), intent(in) :: params
       real(c_double), value :: alpha
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_autotune_sw_f64(params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_autotune_sw_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     ! the GOcean `shallow` kernels one by one (one launch per PSy loop nest); arrays in the kernels' own order
     function dlesm_compute_cu_f64(offset, ld, ny, xstart, xstop, ystart, ystop, cu, p, u, stream) &
          bind(C, name="dlesm_compute_cu_f64") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: cu, p, u, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_cv_f64(offset, ld, ny, xstart, xstop, ystart, ystop, cv, p, v, stream) &
          bind(C, name="dlesm_compute_cv_f64") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: cv, p, v, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_z_f64(offset, ld, ny, xstart, xstop, ystart, ystop, fsdx, fsdy, z, p, u, v, stream) &
          bind(C, name="dlesm_compute_z_f64") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: fsdx, fsdy
       type(c_ptr), value :: z, p, u, v, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_h_f64(offset, ld, ny, xstart, xstop, ystart, ystop, h, p, u, v, stream) &
          bind(C, name="dlesm_compute_h_f64") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: h, p, u, v, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_unew_f64(offset, ld, ny, xstart, xstop, ystart, ystop, tdts8, tdtsdx, unew, uold, z, cv, &
          h, stream) bind(C, name="dlesm_compute_unew_f64") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: tdts8, tdtsdx
       type(c_ptr), value :: unew, uold, z, cv, h, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_vnew_f64(offset, ld, ny, xstart, xstop, ystart, ystop, tdts8, tdtsdy, vnew, vold, z, cu, &
          h, stream) bind(C, name="dlesm_compute_vnew_f64") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: tdts8, tdtsdy
       type(c_ptr), value :: vnew, vold, z, cu, h, stream
       integer(c_int) :: rc
     end function
     function dlesm_compute_pnew_f64(offset, ld, ny, xstart, xstop, ystart, ystop, tdtsdx, tdtsdy, pnew, pold, cu, cv, &
          stream) bind(C, name="dlesm_compute_pnew_f64") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: offset, ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: tdtsdx, tdtsdy
       type(c_ptr), value :: pnew, pold, cu, cv, stream
       integer(c_int) :: rc
     end function
     function dlesm_time_smooth_f64(ld, ny, xstart, xstop, ystart, ystop, alpha, field, field_new, field_old, stream) &
          bind(C, name="dlesm_time_smooth_f64") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: alpha
       type(c_ptr), value :: field, field_new, field_old, stream
       integer(c_int) :: rc
     end function
     function dlesm_periodic_halos_apply_multi_f64(fields, nfields, ld, ny, internal, bc_x, bc_y, stream) &
          bind(C, name="dlesm_periodic_halos_apply_multi_f64") result(rc)
       import :: c_int, c_ptr, c_region
       type(c_ptr), intent(in) :: fields(*)
       integer(c_int), value :: nfields, ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       type(c_ptr), value :: stream
       integer(c_int) :: rc
     end function
     function dlesm_periodic_halos_apply_f64(field, ld, ny, internal, bc_x, bc_y, stream) &
          bind(C, name="dlesm_periodic_halos_apply_f64") result(rc)
       import :: c_int, c_ptr, c_region
       type(c_ptr), value :: field, stream
       integer(c_int), value :: ld, ny, bc_x, bc_y
       type(c_region), intent(in) :: internal
       integer(c_int) :: rc
     end function
     function dlesm_shallow_autotune_f64(params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_autotune_f64") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_copy_patch_f64(src, dst, ld, ny_arr, sx0, sy0, dx0, dy0, nx, ny, stream) &
          bind(C, name="dlesm_copy_patch_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: src, dst, stream
       integer(c_int), value :: ld, ny_arr, sx0, sy0, dx0, dy0, nx, ny
       integer(c_int) :: rc
     end function
     function dlesm_fill_f64(f, ld, ny, xstart, xstop, ystart, ystop, val, stream) &
          bind(C, name="dlesm_fill_f64") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: f, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       real(c_double), value :: val
       integer(c_int) :: rc
     end function
     function dlesm_checksum_f64(f, ld, ny, xstart, xstop, ystart, ystop, res, stream) &
          bind(C, name="dlesm_checksum_f64") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: f, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       real(c_double), intent(out) :: res
       integer(c_int) :: rc
     end function
     function dlesm_hash_init_f64(f, ld, ny, xstart, xstop, ystart, ystop, seed, gx0, gy0, stream) &
          bind(C, name="dlesm_hash_init_f64") result(rc)
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: f, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int64_t), value :: seed, gx0, gy0
       integer(c_int) :: rc
     end function
     ! ---- halo exchange ------------------------------------------------------
     function dlesm_comm_unique_id(id) bind(C, name="dlesm_comm_unique_id") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(*)
       integer(c_int) :: rc
     end function
     function dlesm_rendezvous_remove(path) bind(C, name="dlesm_rendezvous_remove") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int) :: rc
     end function
     function dlesm_rendezvous_publish(path, id, token) bind(C, name="dlesm_rendezvous_publish") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*), id(*), token(*)
       integer(c_int) :: rc
     end function
     function dlesm_rendezvous_fetch(path, id, token, timeout_ms) &
          bind(C, name="dlesm_rendezvous_fetch") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*), token(*)
       character(kind=c_char), intent(out) :: id(*)
       integer(c_int), value :: timeout_ms
       integer(c_int) :: rc
     end function
     function dlesm_rendezvous_ack(path, rank0) bind(C, name="dlesm_rendezvous_ack") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int), value :: rank0
       integer(c_int) :: rc
     end function
     function dlesm_rendezvous_wait_acks(path, nranks, timeout_ms) &
          bind(C, name="dlesm_rendezvous_wait_acks") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int), value :: nranks, timeout_ms
       integer(c_int) :: rc
     end function
     function dlesm_comm_init(id, nranks, rank0) bind(C, name="dlesm_comm_init") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: id(*)
       integer(c_int), value :: nranks, rank0
       integer(c_int) :: rc
     end function
     ! mailbox mode: no RCCL communicator (the session name is made by rank 0 and handed round like an RCCL id)
     function dlesm_board_nonce(id) bind(C, name="dlesm_board_nonce") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(*)
       integer(c_int) :: rc
     end function
     function dlesm_board_abort(msg) bind(C, name="dlesm_board_abort") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: msg(*)
       integer(c_int) :: rc
     end function
     function dlesm_comm_init_mailbox(id, nranks, rank0) bind(C, name="dlesm_comm_init_mailbox") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: id(*)
       integer(c_int), value :: nranks, rank0
       integer(c_int) :: rc
     end function
     function dlesm_comm_finalize() bind(C, name="dlesm_comm_finalize") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function
     function dlesm_halo_plan_create(tables, ld, ny, plan) &
          bind(C, name="dlesm_halo_plan_create") result(rc)
       import :: c_int, c_ptr, c_comm_tables
       type(c_comm_tables), intent(in) :: tables
       integer(c_int), value :: ld, ny
       type(c_ptr), intent(out) :: plan
       integer(c_int) :: rc
     end function
     function dlesm_halo_plan_destroy(plan) bind(C, name="dlesm_halo_plan_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan
       integer(c_int) :: rc
     end function
     function dlesm_halo_exchange_f64(plan, field, dirs_mask, stream) &
          bind(C, name="dlesm_halo_exchange_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, field, stream
       integer(c_int), value :: dirs_mask
       integer(c_int) :: rc
     end function
     function dlesm_jacobi5_step_dm(plan, in, out, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_jacobi5_step_dm") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, in, out, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_stencil5_autotune_f64(in, out, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_stencil5_autotune_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: in, out, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_stencil5_multi_f64(in, out, ld, ny, nsteps, xstart, xstop, ystart, ystop, &
          exstart, exstop, eystart, eystop, grow_w, grow_e, grow_s, grow_n, stream) &
          bind(C, name="dlesm_stencil5_multi_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: in, out, stream
       integer(c_int), value :: ld, ny, nsteps, xstart, xstop, ystart, ystop
       integer(c_int), value :: exstart, exstop, eystart, eystop, grow_w, grow_e, grow_s, grow_n
       integer(c_int) :: rc
     end function
     function dlesm_jacobi5_step_dm_pipelined(plan, in, out, ld, ny, xstart, xstop, ystart, ystop, stream) &
          bind(C, name="dlesm_jacobi5_step_dm_pipelined") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, in, out, stream
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_halo_plan_join(plan, stream) bind(C, name="dlesm_halo_plan_join") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, stream
       integer(c_int) :: rc
     end function
     ! peer transport: connect the plan to the neighbours' mailboxes (collective; the blobs travel by ncclAllGather)
     function dlesm_halo_plan_peer_connect_rccl(plan, nfields) bind(C, name="dlesm_halo_plan_peer_connect_rccl") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan
       integer(c_int), value :: nfields
       integer(c_int) :: rc
     end function
     function dlesm_halo_plan_peer_connected(plan) bind(C, name="dlesm_halo_plan_peer_connected") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan
       integer(c_int) :: rc
     end function
     function dlesm_jacobi5_multi_step_dm(plan, in, out, ld, ny, nsteps, xstart, xstop, ystart, ystop, &
          stream) bind(C, name="dlesm_jacobi5_multi_step_dm") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, in, out, stream
       integer(c_int), value :: ld, ny, nsteps, xstart, xstop, ystart, ystop
       integer(c_int) :: rc
     end function
     function dlesm_halo_exchange_multi_f64(plan, fields, nfields, dirs_mask, stream) &
          bind(C, name="dlesm_halo_exchange_multi_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan, stream
       type(c_ptr), intent(in) :: fields(*)
       integer(c_int), value :: nfields, dirs_mask
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_dm(plan, params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_dm") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_ptr), value :: plan
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_shallow_step_dm_pipelined(plan, params, ld, ny, xstart, xstop, ystart, ystop, u, v, p, &
          uold, vold, pold, unew, vnew, pnew, stream) bind(C, name="dlesm_shallow_step_dm_pipelined") result(rc)
       import :: c_int, c_ptr, c_sw_params
       type(c_ptr), value :: plan
       type(c_sw_params), intent(in) :: params
       integer(c_int), value :: ld, ny, xstart, xstop, ystart, ystop
       type(c_ptr), value :: u, v, p, uold, vold, pold, unew, vnew, pnew, stream
       integer(c_int) :: rc
     end function
     function dlesm_global_sum_f64(val) bind(C, name="dlesm_global_sum_f64") result(rc)
       import :: c_int, c_double
       real(c_double), intent(inout) :: val
       integer(c_int) :: rc
     end function
     function dlesm_gather_inner_f64(field, ld, ny, internal, decomp, subdomains, nranks, global_host) &
          bind(C, name="dlesm_gather_inner_f64") result(rc)
       import :: c_int, c_ptr, c_region, c_decomp, c_subdomain
       type(c_ptr), value :: field, global_host
       integer(c_int), value :: ld, ny, nranks
       type(c_region), intent(in) :: internal
       type(c_decomp), intent(in) :: decomp
       type(c_subdomain), intent(in) :: subdomains(*)
       integer(c_int) :: rc
     end function
     function dlesm_gather_f64(send, recv, n) bind(C, name="dlesm_gather_f64") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: send, recv
       integer(c_int), value :: n
       integer(c_int) :: rc
     end function
     ! ---- libc / HIP helpers the Fortran layer needs ---------------------------
     function c_strlen(s) bind(C, name="strlen") result(n)
       import :: c_ptr, c_size_t
       type(c_ptr), value :: s
       integer(c_size_t) :: n
     end function
     function hipDeviceSynchronize() bind(C, name="hipDeviceSynchronize") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function
     function hipMalloc(ptr, nbytes) bind(C, name="hipMalloc") result(rc)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), intent(out) :: ptr
       integer(c_size_t), value :: nbytes
       integer(c_int) :: rc
     end function
     function hipFree(ptr) bind(C, name="hipFree") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ptr
       integer(c_int) :: rc
     end function
     function hipMemcpy(dst, src, nbytes, kind) bind(C, name="hipMemcpy") result(rc)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: dst, src
       integer(c_size_t), value :: nbytes
       integer(c_int), value :: kind
       integer(c_int) :: rc
     end function
  end interface

contains

  !> Text of the library's last error as a Fortran string
  function dlesm_error_text() result(txt)
    character(len=:), allocatable :: txt
    type(c_ptr) :: p
    character(kind=c_char), pointer :: chars(:)
    integer :: n, i
    p = dlesm_last_error()
    if (.not. c_associated(p)) then
       txt = ''
       return
    end if
    n = int(c_strlen(p))
    call c_f_pointer(p, chars, [n])
    allocate(character(len=n) :: txt)
    do i = 1, n
       txt(i:i) = chars(i)
    end do
  end function dlesm_error_text

end module dlesm_hip_mod
