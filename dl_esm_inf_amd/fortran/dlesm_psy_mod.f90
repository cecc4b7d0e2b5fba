!> PSy-layer building blocks for HIP: what replaces the generated
!!     do jj = fld%internal%ystart, fld%internal%ystop
!!        do ji = fld%internal%xstart, fld%internal%xstop
!!           call kern_code(ji, jj, out%data, in%data, ...)
!! loop nests (form: reference infrastructure_mod.f90:32-41) -- one call that launches the
!! matching CDNA4 kernel over the same index box on the fields' device copies.
!!
!! Also provides infrastructure_mod, the reference's field-copy "kernel" module, with its
!! metadata type and pointwise code unchanged in meaning.
module infrastructure_mod
  use kind_params_mod
  use kernel_mod
  use argument_mod
  use grid_mod
  use field_mod
  implicit none

  type, extends(kernel_type) :: copy
     type(go_arg), dimension(2) :: meta_args = &
          (/ go_arg(GO_WRITE, GO_EVERY, GO_POINTWISE), &
             go_arg(GO_READ,  GO_EVERY, GO_POINTWISE) /)
     integer :: ITERATES_OVER = GO_ALL_PTS
     integer :: index_offset = GO_OFFSET_ANY
   contains
     procedure, nopass :: code => field_copy_code
  end type copy

contains

  subroutine field_copy_code(ji, jj, output, input)
    integer, intent(in) :: ji, jj
    real(go_wp), dimension(:,:), intent(in) :: input
    real(go_wp), dimension(:,:), intent(out) :: output
    output(ji, jj) = input(ji, jj)
  end subroutine field_copy_code

end module infrastructure_mod


module dlesm_psy_mod
  use iso_c_binding
  use kind_params_mod
  use grid_mod
  use field_mod
  use gocean_mod, only: gocean_stop
  use dlesm_hip_mod
  implicit none
  private

  public :: invoke_jacobi5_masked, invoke_jacobi5_dm_pipelined, halo_join, halo_connect_peers
  public :: invoke_shallow_step_sw, invoke_periodic_halos, invoke_stencil9, invoke_stencil9_dm
  public :: invoke_jacobi5, invoke_jacobi5_dm, invoke_shallow_step, invoke_copy, invoke_hash_init
  public :: invoke_shallow_step_dm_pipelined, invoke_continuity
  public :: invoke_shallow_step_dm, halo_exchange_multi, invoke_jacobi5_multi, plan_jacobi5, plan_shallow_step
  public :: shallow_params, c_sw_params, device_sync, grid_to_device
  public :: invoke_compute_cu, invoke_compute_cv, invoke_compute_z, invoke_compute_h
  public :: invoke_compute_unew, invoke_compute_vnew, invoke_compute_pnew, invoke_time_smooth
  public :: invoke_shallow_step_sw_periodic, plan_shallow_step_sw, invoke_periodic_halos_multi
  public :: invoke_shallow_step_smooth, invoke_shallow_step_sw_smooth_periodic, invoke_shallow_step_smooth_dm
  public :: invoke_shallow_step_x2, invoke_shallow_step_smooth_x2
  public :: invoke_shallow_step_sw_x2_periodic, invoke_shallow_step_sw_smooth_x2_periodic

contains

  subroutine device_sync()
    if (hipDeviceSynchronize() /= 0) call gocean_stop('device synchronisation failed')
  end subroutine device_sync

  subroutine need_device(f)
    type(r2d_field), intent(inout), target :: f
    if (.not. f%data_on_device) call field_to_device(f)
    if (.not. field_on_dlesm_device(f)) call gocean_stop('PSy layer: field lives on a foreign device')
  end subroutine need_device

  !> out = 0.25*((w+e)+(s+n)) of `in` over out%internal
  subroutine invoke_jacobi5(out, in)
    type(r2d_field), intent(inout), target :: out, in
    integer(c_int) :: rc
    call need_device(in);  call need_device(out)
    rc = dlesm_stencil5_f64(field_device_data(in), field_device_data(out), &
                            int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                            int(out%internal%xstart, c_int), int(out%internal%xstop, c_int), &
                            int(out%internal%ystart, c_int), int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_jacobi5: ' // dlesm_error_text())
  end subroutine invoke_jacobi5

  !> A general 3x3 weighted stencil: out(ji,jj) = SUM coef(di,dj)*in(ji+di,jj+dj) over out%internal --
  !! the PSy layer of a kernel with a GO_STENCIL(111,111,111) read argument and nine real scalars.
  subroutine invoke_stencil9(out, in, coef)
    type(r2d_field), intent(inout), target :: out, in
    real(go_wp), intent(in) :: coef(-1:1, -1:1)
    real(c_double) :: c9(9)
    integer(c_int) :: rc
    integer :: di, dj
    call need_device(in);  call need_device(out)
    ! di fastest, south row first: the C order coef[(dj+1)*3 + (di+1)].  (Explicit loops: amdflang 22 -O2
    ! turns reshape() of an array with lower bounds -1 into a broadcast of its first element.)
    do dj = -1, 1
       do di = -1, 1
          c9(3 * (dj + 1) + di + 2) = coef(di, dj)
       end do
    end do
    rc = dlesm_stencil9_f64(field_device_data(in), field_device_data(out), c9, &
                            int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                            int(out%internal%xstart, c_int), int(out%internal%xstop, c_int), &
                            int(out%internal%ystart, c_int), int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_stencil9: ' // dlesm_error_text())
  end subroutine invoke_stencil9

  !> invoke_stencil9 + out%halo_exchange(1), the exchange hidden behind the interior sweep
  !! (all eight directions when a corner weight is non-zero, the four edges otherwise).
  subroutine invoke_stencil9_dm(out, in, coef)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: out, in
    real(go_wp), intent(in) :: coef(-1:1, -1:1)
    real(c_double) :: c9(9)
    integer(c_int) :: rc
    integer :: di, dj
    if (.not. DIST_MEM_ENABLED) then
       call invoke_stencil9(out, in, coef)
       return
    end if
    call need_device(in);  call need_device(out)
    do dj = -1, 1
       do di = -1, 1
          c9(3 * (dj + 1) + di + 2) = coef(di, dj)
       end do
    end do
    rc = dlesm_stencil9_step_dm(halo_plan_for(out%grid%nx, out%grid%ny), field_device_data(in), &
                                field_device_data(out), c9, int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                                int(out%internal%xstart, c_int), int(out%internal%xstop, c_int), &
                                int(out%internal%ystart, c_int), int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_stencil9_dm: ' // dlesm_error_text())
  end subroutine invoke_stencil9_dm

  !> The PSy layer of a kernel whose metadata requests the T mask,
  !!   go_arg(GO_WRITE, GO_CT, GO_POINTWISE), go_arg(GO_READ, GO_CT, GO_STENCIL(010,111,010)),
  !!   go_arg(GO_READ, GO_GRID_MASK_T)                          (argument_mod.f90:75-112)
  !! i.e. `call jacobi5_masked_code(ji, jj, out%data, in%data, out%grid%tmask)` over out%internal:
  !! the kernel gets the grid's mask -- on the device its mirror grid%tmask_device, created on
  !! first use.  Dry points carry their value over, dry neighbours are mirrored.
  subroutine invoke_jacobi5_masked(out, in)
    type(r2d_field), intent(inout), target :: out, in
    integer(c_int) :: rc
    call need_device(in);  call need_device(out)
    if (.not. c_associated(out%grid%tmask_device)) call grid_to_device(out%grid)
    rc = dlesm_stencil5_masked_f64(field_device_data(in), field_device_data(out), out%grid%tmask_device, &
                                   int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                                   int(out%internal%xstart, c_int), int(out%internal%xstop, c_int), &
                                   int(out%internal%ystart, c_int), int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_jacobi5_masked: ' // dlesm_error_text())
  end subroutine invoke_jacobi5_masked

  !> The PSy layer of a kernel on all three point types whose metadata requests the cell area,
  !!   go_arg(GO_WRITE, GO_CT, GO_POINTWISE), go_arg(GO_READ, GO_CT, GO_POINTWISE),
  !!   go_arg(GO_READ, GO_CU, GO_STENCIL(000,110,000)) x 3, go_arg(GO_READ, GO_CV, GO_STENCIL(000,010,010)) x 3,
  !!   go_arg(GO_READ, GO_R_SCALAR, GO_POINTWISE), go_arg(GO_READ, GO_GRID_AREA_T)
  !! i.e. `call continuity_code(ji, jj, ssha%data, sshn_t%data, sshn_u%data, sshn_v%data, hu%data, hv%data,
  !! un%data, vn%data, rdt, ssha%grid%area_t)` over ssha%internal (the free-surface update of a
  !! NEMOLite2D-class model): the kernel gets the grid's area_t, on the device its mirror.
  subroutine invoke_continuity(ssha, sshn_t, sshn_u, sshn_v, hu, hv, un, vn, rdt)
    type(r2d_field), intent(inout), target :: ssha, sshn_t, sshn_u, sshn_v, hu, hv, un, vn
    real(go_wp), intent(in) :: rdt
    integer(c_int) :: rc
    call need_device(ssha);  call need_device(sshn_t);  call need_device(sshn_u);  call need_device(sshn_v)
    call need_device(hu);  call need_device(hv);  call need_device(un);  call need_device(vn)
    if (.not. c_associated(ssha%grid%area_t_device)) call grid_to_device(ssha%grid)
    rc = dlesm_continuity_f64(real(rdt, c_double), int(ssha%grid%nx, c_int), int(ssha%grid%ny, c_int), &
                              int(ssha%internal%xstart, c_int), int(ssha%internal%xstop, c_int), &
                              int(ssha%internal%ystart, c_int), int(ssha%internal%ystop, c_int), &
                              field_device_data(sshn_t), field_device_data(sshn_u), field_device_data(sshn_v), &
                              field_device_data(hu), field_device_data(hv), field_device_data(un), &
                              field_device_data(vn), ssha%grid%area_t_device, field_device_data(ssha), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_continuity: ' // dlesm_error_text())
  end subroutine invoke_continuity

  !> Optional planning call (once per field geometry, outside the time loop): lets the library time
  !! its launch shapes for invoke_jacobi5 / invoke_jacobi5_dm on these fields and keep the fastest.
  !! `out` receives one valid step of `in`; results never depend on the shape.
  subroutine plan_jacobi5(out, in)
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: out, in
    integer(c_int) :: rc, shrink
    call need_device(in);  call need_device(out)
    shrink = 0
    if (DIST_MEM_ENABLED) shrink = 1         ! the distributed step's interior box
    rc = dlesm_stencil5_autotune_f64(field_device_data(in), field_device_data(out), &
                                     int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                                     int(out%internal%xstart + shrink, c_int), int(out%internal%xstop - shrink, c_int), &
                                     int(out%internal%ystart + shrink, c_int), int(out%internal%ystop - shrink, c_int), &
                                     c_null_ptr)
    if (rc /= 0) call gocean_stop('plan_jacobi5: ' // dlesm_error_text())
  end subroutine plan_jacobi5

  !> Distributed Jacobi step: `in` must have valid halos; on return (asynchronously) `out`
  !! holds the update AND its halos, the exchange having run behind the interior sweep.
  subroutine invoke_jacobi5_dm(out, in)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: out, in
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) then
       call invoke_jacobi5(out, in)
       return
    end if
    call need_device(in);  call need_device(out)
    rc = dlesm_jacobi5_step_dm(halo_plan_for(out%grid%nx, out%grid%ny), field_device_data(in), &
                               field_device_data(out), int(out%grid%nx, c_int), int(out%grid%ny, c_int), &
                               int(out%internal%xstart, c_int), int(out%internal%xstop, c_int), &
                               int(out%internal%ystart, c_int), int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_jacobi5_dm: ' // dlesm_error_text())
  end subroutine invoke_jacobi5_dm

  !> The same step for a time loop of such steps: returns with the exchange of `out` in flight;
  !! the next invoke_jacobi5_dm_pipelined on this grid waits for it on the device, any other library
  !! call on the grid's plan joins it first, and halo_join(grid) orders everything else behind it.
  subroutine invoke_jacobi5_dm_pipelined(out, in)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: out, in
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) then
       call invoke_jacobi5(out, in)
       return
    end if
    call need_device(in);  call need_device(out)
    rc = dlesm_jacobi5_step_dm_pipelined(halo_plan_for(out%grid%nx, out%grid%ny), field_device_data(in), &
                                         field_device_data(out), int(out%grid%nx, c_int), &
                                         int(out%grid%ny, c_int), int(out%internal%xstart, c_int), &
                                         int(out%internal%xstop, c_int), int(out%internal%ystart, c_int), &
                                         int(out%internal%ystop, c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_jacobi5_dm_pipelined: ' // dlesm_error_text())
  end subroutine invoke_jacobi5_dm_pipelined

  !> COLLECTIVE, once per grid: connect the grid's message plan to the neighbours' mailboxes.  From then on
  !! invoke_jacobi5_dm / invoke_jacobi5_dm_pipelined exchange by storing straight into the neighbours' memory over
  !! xGMI (the frame workgroups of the step launch are the exchange) instead of through an RCCL group per step --
  !! what MPI_Isend/Irecv/Waitany do per strip in the reference (parallel_comms_mod.f90:1601-1750).  Same results.
  !! nfields (default 1): 3 also takes the distributed shallow-water steps through the mailboxes.
  subroutine halo_connect_peers(grid, nfields)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(grid_type), intent(in) :: grid
    integer, intent(in), optional :: nfields
    integer(c_int) :: rc, nf
    if (.not. DIST_MEM_ENABLED) return
    if (dlesm_halo_plan_peer_connected(halo_plan_for(grid%nx, grid%ny)) /= 0) return
    nf = 1
    if (present(nfields)) nf = int(nfields, c_int)
    rc = dlesm_halo_plan_peer_connect_rccl(halo_plan_for(grid%nx, grid%ny), nf)
    if (rc /= 0) call gocean_stop('halo_connect_peers: ' // dlesm_error_text())
  end subroutine halo_connect_peers

  subroutine halo_join(grid)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(grid_type), intent(in) :: grid
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) return
    rc = dlesm_halo_plan_join(halo_plan_for(grid%nx, grid%ny), c_null_ptr)
    if (rc /= 0) call gocean_stop('halo_join: ' // dlesm_error_text())
  end subroutine halo_join

  !> nsteps (2..8) Jacobi time steps in one sweep (temporal blocking).  Serial / one tile: the
  !! boundary ring of `in` stays fixed through all steps.  Distributed (grid decomposed with
  !! halo_width = nsteps): one depth-nsteps halo exchange per call, hidden behind the interior;
  !! `in` must hold valid depth-nsteps halos and `out` leaves with them.  Bit-identical to nsteps
  !! calls of invoke_jacobi5 (+ halo exchanges).
  subroutine invoke_jacobi5_multi(out, in, nsteps)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: out, in
    integer, intent(in) :: nsteps
    integer(c_int) :: rc
    call need_device(in);  call need_device(out)
    associate (b => out%internal)
      if (DIST_MEM_ENABLED) then
         rc = dlesm_jacobi5_multi_step_dm(halo_plan_for(out%grid%nx, out%grid%ny), field_device_data(in), &
                                          field_device_data(out), int(out%grid%nx, c_int), &
                                          int(out%grid%ny, c_int), int(nsteps, c_int), int(b%xstart, c_int), &
                                          int(b%xstop, c_int), int(b%ystart, c_int), int(b%ystop, c_int), &
                                          c_null_ptr)
      else
         rc = dlesm_stencil5_multi_f64(field_device_data(in), field_device_data(out), &
                                       int(out%grid%nx, c_int), int(out%grid%ny, c_int), int(nsteps, c_int), &
                                       int(b%xstart, c_int), int(b%xstop, c_int), int(b%ystart, c_int), &
                                       int(b%ystop, c_int), int(b%xstart, c_int), int(b%xstop, c_int), &
                                       int(b%ystart, c_int), int(b%ystop, c_int), 0_c_int, 0_c_int, 0_c_int, &
                                       0_c_int, c_null_ptr)
      end if
    end associate
    if (rc /= 0) call gocean_stop('invoke_jacobi5_multi: ' // dlesm_error_text())
  end subroutine invoke_jacobi5_multi

  !> Constants of the shallow-water step; tdt = 2*dt (leapfrog)
  function shallow_params(dx, dy, dt) result(p)
    real(go_wp), intent(in) :: dx, dy, dt
    type(c_sw_params) :: p
    real(go_wp) :: tdt
    tdt = dt + dt
    p%fsdx = 4.0_go_wp / dx;  p%fsdy = 4.0_go_wp / dy
    p%tdts8 = tdt / 8.0_go_wp
    p%tdtsdx = tdt / dx;  p%tdtsdy = tdt / dy
  end function shallow_params

  subroutine invoke_shallow_step(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_step_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                field_device_data(u), field_device_data(v), field_device_data(p), &
                                field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step: ' // dlesm_error_text())
  end subroutine invoke_shallow_step

  !> The SW-offset form (the staggering of the GOcean `shallow` benchmark; with periodic boundaries the
  !! only configuration the reference supports for it, serially).  u, v, p must hold valid periodic
  !! halos; follow it with invoke_periodic_halos on the three new fields.
  subroutine invoke_shallow_step_sw(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_step_sw_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                   int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                   int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                   field_device_data(u), field_device_data(v), field_device_data(p), &
                                   field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                   field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                   c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_sw: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_sw

  !> invoke_shallow_step_sw over the internal region AND the periodic copies of the three new fields, in ONE launch:
  !! the step's edge tiles store the periodic images themselves (== invoke_shallow_step_sw followed by
  !! invoke_periodic_halos of unew, vnew, pnew, bit for bit)
  subroutine invoke_shallow_step_sw_periodic(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    type(c_region) :: cint
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    associate (it => p%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_shallow_step_sw_periodic_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), cint, &
                                            int(p%grid%boundary_conditions(1), c_int), &
                                            int(p%grid%boundary_conditions(2), c_int), &
                                            field_device_data(u), field_device_data(v), field_device_data(p), &
                                            field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                            field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                            c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_sw_periodic: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_sw_periodic

  !> One WHOLE time step of the GOcean leapfrog in one launch (NE offset): the u/v/h update and the Asselin filter of the
  !! old level (time_smooth) in place -- == invoke_shallow_step followed by invoke_time_smooth of u, v and p, bit for bit, at
  !! 96 B/cell instead of 168.  Afterwards rotate u <- unew (uold already holds the filtered u).
  subroutine invoke_shallow_step_smooth(prm, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    real(go_wp), intent(in) :: alpha
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_step_smooth_f64(prm, alpha, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                       int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                       int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                       field_device_data(u), field_device_data(v), field_device_data(p), &
                                       field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                       field_device_data(unew), field_device_data(vnew), field_device_data(pnew), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_smooth: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_smooth

  !> TWO leapfrog steps in one launch (NE offset, fixed boundary ring): level n+1 into unew, vnew, pnew and level n+2 into
  !! unew2, vnew2, pnew2 -- == invoke_shallow_step(prm, u, v, p, uold, vold, pold, unew, vnew, pnew) followed by
  !! invoke_shallow_step(prm, unew, vnew, pnew, u, v, p, unew2, vnew2, pnew2), bit for bit, at 48 instead of 72 B/cell/step.
  !! Twelve distinct fields; afterwards rotate (cur, old, new1, new2) <- (new2, new1, old, cur).
  subroutine invoke_shallow_step_x2(prm, u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    call need_device(unew2);  call need_device(vnew2);  call need_device(pnew2)
    rc = dlesm_shallow_step_x2_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                   int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                   int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                   field_device_data(u), field_device_data(v), field_device_data(p), &
                                   field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                   field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                   field_device_data(unew2), field_device_data(vnew2), field_device_data(pnew2), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_x2: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_x2

  !> TWO whole time steps of the GOcean leapfrog (update + Asselin filter of the old level, twice) in one launch (NE offset):
  !! level n+2 into unew2, vnew2, pnew2 and the filtered level n+1 into uold2, vold2, pold2; u .. pold are not modified.
  !! == two invoke_shallow_step_smooth calls with the usual rotation, bit for bit, at 48 instead of 96 B/cell/step.
  !! Time loop: ping-pong (u, v, p, uold, vold, pold) <-> (unew2, vnew2, pnew2, uold2, vold2, pold2).
  subroutine invoke_shallow_step_smooth_x2(prm, alpha, u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2)
    type(c_sw_params), intent(in) :: prm
    real(go_wp), intent(in) :: alpha
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew2);  call need_device(vnew2);  call need_device(pnew2)
    call need_device(uold2);  call need_device(vold2);  call need_device(pold2)
    rc = dlesm_shallow_step_smooth_x2_f64(prm, alpha, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                          int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                          int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                          field_device_data(u), field_device_data(v), field_device_data(p), &
                                          field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                          field_device_data(unew2), field_device_data(vnew2), field_device_data(pnew2), &
                                          field_device_data(uold2), field_device_data(vold2), field_device_data(pold2), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_smooth_x2: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_smooth_x2

  !> Two steps of the SW-offset periodic model in one launch (== two invoke_shallow_step_sw_periodic calls): level n+1 with its
  !! periodic images into unew .. pnew, level n+2 with its images into unew2 .. pnew2.
  subroutine invoke_shallow_step_sw_x2_periodic(prm, u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2
    type(c_region) :: cint
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    call need_device(unew2);  call need_device(vnew2);  call need_device(pnew2)
    associate (it => p%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_shallow_step_sw_x2_periodic_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), cint, &
                                               int(p%grid%boundary_conditions(1), c_int), int(p%grid%boundary_conditions(2), c_int), &
                                               field_device_data(u), field_device_data(v), field_device_data(p), &
                                               field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                               field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                               field_device_data(unew2), field_device_data(vnew2), field_device_data(pnew2), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_sw_x2_periodic: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_sw_x2_periodic

  !> TWO whole time steps of the GOcean `shallow` benchmark in one launch: update, Asselin filter of the old level and periodic
  !! images, twice.  Level n+2 into unew2 .. pnew2, the filtered level n+1 into uold2 .. pold2 (both with their images); u .. pold
  !! are not modified.  == two invoke_shallow_step_sw_smooth_periodic calls with the loop's rotation, at 48 instead of 96 B/cell/step.
  subroutine invoke_shallow_step_sw_smooth_x2_periodic(prm, alpha, u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2)
    type(c_sw_params), intent(in) :: prm
    real(go_wp), intent(in) :: alpha
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2
    type(c_region) :: cint
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew2);  call need_device(vnew2);  call need_device(pnew2)
    call need_device(uold2);  call need_device(vold2);  call need_device(pold2)
    associate (it => p%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_shallow_step_sw_smooth_x2_periodic_f64(prm, alpha, int(p%grid%nx, c_int), int(p%grid%ny, c_int), cint, &
                                                      int(p%grid%boundary_conditions(1), c_int), &
                                                      int(p%grid%boundary_conditions(2), c_int), &
                                                      field_device_data(u), field_device_data(v), field_device_data(p), &
                                                      field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                                      field_device_data(unew2), field_device_data(vnew2), field_device_data(pnew2), &
                                                      field_device_data(uold2), field_device_data(vold2), field_device_data(pold2), &
                                                      c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_sw_smooth_x2_periodic: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_sw_smooth_x2_periodic

  !> The same for the SW-offset periodic model: update, filter and the periodic images of the new and of the filtered old
  !! level in ONE launch -- a whole time step of the GOcean `shallow` benchmark.
  subroutine invoke_shallow_step_sw_smooth_periodic(prm, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    real(go_wp), intent(in) :: alpha
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    type(c_region) :: cint
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    associate (it => p%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_shallow_step_sw_smooth_periodic_f64(prm, alpha, int(p%grid%nx, c_int), int(p%grid%ny, c_int), cint, &
                                                   int(p%grid%boundary_conditions(1), c_int), &
                                                   int(p%grid%boundary_conditions(2), c_int), &
                                                   field_device_data(u), field_device_data(v), field_device_data(p), &
                                                   field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                                   field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                                   c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_sw_smooth_periodic: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_sw_smooth_periodic

  !> The distributed form of invoke_shallow_step_smooth (one launch + the exchange of the new level hidden behind the interior):
  !! `pipelined` = the time-loop form (halo_join(grid) after the loop); serial builds fall back to invoke_shallow_step_smooth.
  subroutine invoke_shallow_step_smooth_dm(prm, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew, pipelined)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(c_sw_params), intent(in) :: prm
    real(go_wp), intent(in) :: alpha
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    logical, intent(in), optional :: pipelined
    logical :: pipe
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) then
       call invoke_shallow_step_smooth(prm, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew)
       return
    end if
    pipe = .false.
    if (present(pipelined)) pipe = pipelined
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    if (pipe) then
       rc = dlesm_shallow_step_smooth_dm_pipelined(halo_plan_for(p%grid%nx, p%grid%ny), prm, alpha, &
                int(p%grid%nx, c_int), int(p%grid%ny, c_int), int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                field_device_data(u), field_device_data(v), field_device_data(p), &
                field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                field_device_data(unew), field_device_data(vnew), field_device_data(pnew), c_null_ptr)
    else
       rc = dlesm_shallow_step_smooth_dm(halo_plan_for(p%grid%nx, p%grid%ny), prm, alpha, &
                int(p%grid%nx, c_int), int(p%grid%ny, c_int), int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                field_device_data(u), field_device_data(v), field_device_data(p), &
                field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                field_device_data(unew), field_device_data(vnew), field_device_data(pnew), c_null_ptr)
    end if
    if (rc /= 0) call gocean_stop('invoke_shallow_step_smooth_dm: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_smooth_dm

  !> plan_shallow_step for the SW-offset step
  subroutine plan_shallow_step_sw(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_autotune_sw_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                       int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                       int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                       field_device_data(u), field_device_data(v), field_device_data(p), &
                                       field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                       field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                       c_null_ptr)
    if (rc /= 0) call gocean_stop('plan_shallow_step_sw: ' // dlesm_error_text())
  end subroutine plan_shallow_step_sw

  ! ---- The GOcean `shallow` kernels ONE BY ONE: what replaces each generated loop nest
  !        do jj = fld%internal%ystart, fld%internal%ystop
  !           do ji = fld%internal%xstart, fld%internal%xstop
  !              call compute_cu_code(ji, jj, cu%data, p%data, u%data)
  !      of an unmodified PSy layer (kernel form: reference infrastructure_mod.f90:13-41; metadata
  !      argument_mod.f90:39-112, kernel_mod.f90:28-50).  Loop bounds: the written field's internal region
  !      (ITERATES_OVER = GO_INTERNAL_PTS) unless `box` = (/xstart, xstop, ystart, ystop/) is given; the kernel's
  !      index_offset is the grid's; dx, dy (the kernels' GO_GRID_DX_CONST / GO_GRID_DY_CONST arguments) come from
  !      the grid.  Formulas: DESIGN.md sections 6, 6.2, 6.3.

  subroutine kernel_box(fld, box, b)
    type(r2d_field), intent(in) :: fld
    integer, intent(in), optional :: box(4)
    integer(c_int), intent(out) :: b(4)
    if (present(box)) then
       b = int(box, c_int)
    else
       b = int((/ fld%internal%xstart, fld%internal%xstop, fld%internal%ystart, fld%internal%ystop /), c_int)
    end if
  end subroutine kernel_box

  subroutine invoke_compute_cu(cu, p, u, box)
    type(r2d_field), intent(inout), target :: cu, p, u
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(cu);  call need_device(p);  call need_device(u)
    call kernel_box(cu, box, b)
    rc = dlesm_compute_cu_f64(int(cu%grid%offset, c_int), int(cu%grid%nx, c_int), int(cu%grid%ny, c_int), &
                              b(1), b(2), b(3), b(4), field_device_data(cu), field_device_data(p), &
                              field_device_data(u), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_cu: ' // dlesm_error_text())
  end subroutine invoke_compute_cu

  subroutine invoke_compute_cv(cv, p, v, box)
    type(r2d_field), intent(inout), target :: cv, p, v
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(cv);  call need_device(p);  call need_device(v)
    call kernel_box(cv, box, b)
    rc = dlesm_compute_cv_f64(int(cv%grid%offset, c_int), int(cv%grid%nx, c_int), int(cv%grid%ny, c_int), &
                              b(1), b(2), b(3), b(4), field_device_data(cv), field_device_data(p), &
                              field_device_data(v), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_cv: ' // dlesm_error_text())
  end subroutine invoke_compute_cv

  subroutine invoke_compute_z(z, p, u, v, box)
    type(r2d_field), intent(inout), target :: z, p, u, v
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(z);  call need_device(p);  call need_device(u);  call need_device(v)
    call kernel_box(z, box, b)
    rc = dlesm_compute_z_f64(int(z%grid%offset, c_int), int(z%grid%nx, c_int), int(z%grid%ny, c_int), &
                             b(1), b(2), b(3), b(4), 4.0_go_wp / z%grid%dx, 4.0_go_wp / z%grid%dy, &
                             field_device_data(z), field_device_data(p), field_device_data(u), field_device_data(v), &
                             c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_z: ' // dlesm_error_text())
  end subroutine invoke_compute_z

  subroutine invoke_compute_h(h, p, u, v, box)
    type(r2d_field), intent(inout), target :: h, p, u, v
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(h);  call need_device(p);  call need_device(u);  call need_device(v)
    call kernel_box(h, box, b)
    rc = dlesm_compute_h_f64(int(h%grid%offset, c_int), int(h%grid%nx, c_int), int(h%grid%ny, c_int), &
                             b(1), b(2), b(3), b(4), field_device_data(h), field_device_data(p), &
                             field_device_data(u), field_device_data(v), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_h: ' // dlesm_error_text())
  end subroutine invoke_compute_h

  !> tdt = 2*dt in a leapfrog step (dt in the first, forward step)
  subroutine invoke_compute_unew(unew, uold, z, cv, h, tdt, box)
    type(r2d_field), intent(inout), target :: unew, uold, z, cv, h
    real(go_wp), intent(in) :: tdt
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(unew);  call need_device(uold);  call need_device(z);  call need_device(cv);  call need_device(h)
    call kernel_box(unew, box, b)
    rc = dlesm_compute_unew_f64(int(unew%grid%offset, c_int), int(unew%grid%nx, c_int), int(unew%grid%ny, c_int), &
                                b(1), b(2), b(3), b(4), tdt / 8.0_go_wp, tdt / unew%grid%dx, &
                                field_device_data(unew), field_device_data(uold), field_device_data(z), &
                                field_device_data(cv), field_device_data(h), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_unew: ' // dlesm_error_text())
  end subroutine invoke_compute_unew

  subroutine invoke_compute_vnew(vnew, vold, z, cu, h, tdt, box)
    type(r2d_field), intent(inout), target :: vnew, vold, z, cu, h
    real(go_wp), intent(in) :: tdt
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(vnew);  call need_device(vold);  call need_device(z);  call need_device(cu);  call need_device(h)
    call kernel_box(vnew, box, b)
    rc = dlesm_compute_vnew_f64(int(vnew%grid%offset, c_int), int(vnew%grid%nx, c_int), int(vnew%grid%ny, c_int), &
                                b(1), b(2), b(3), b(4), tdt / 8.0_go_wp, tdt / vnew%grid%dy, &
                                field_device_data(vnew), field_device_data(vold), field_device_data(z), &
                                field_device_data(cu), field_device_data(h), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_vnew: ' // dlesm_error_text())
  end subroutine invoke_compute_vnew

  subroutine invoke_compute_pnew(pnew, pold, cu, cv, tdt, box)
    type(r2d_field), intent(inout), target :: pnew, pold, cu, cv
    real(go_wp), intent(in) :: tdt
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(pnew);  call need_device(pold);  call need_device(cu);  call need_device(cv)
    call kernel_box(pnew, box, b)
    rc = dlesm_compute_pnew_f64(int(pnew%grid%offset, c_int), int(pnew%grid%nx, c_int), int(pnew%grid%ny, c_int), &
                                b(1), b(2), b(3), b(4), tdt / pnew%grid%dx, tdt / pnew%grid%dy, &
                                field_device_data(pnew), field_device_data(pold), field_device_data(cu), &
                                field_device_data(cv), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_compute_pnew: ' // dlesm_error_text())
  end subroutine invoke_compute_pnew

  !> time_smooth (the Asselin filter of the leapfrog scheme), over field_old%internal:
  !!   field_old = field + alpha*(field_new - 2*field + field_old)
  subroutine invoke_time_smooth(field, field_new, field_old, alpha, box)
    type(r2d_field), intent(inout), target :: field, field_new, field_old
    real(go_wp), intent(in) :: alpha
    integer, intent(in), optional :: box(4)
    integer(c_int) :: rc, b(4)
    call need_device(field);  call need_device(field_new);  call need_device(field_old)
    call kernel_box(field_old, box, b)
    rc = dlesm_time_smooth_f64(int(field_old%grid%nx, c_int), int(field_old%grid%ny, c_int), b(1), b(2), b(3), b(4), &
                               alpha, field_device_data(field), field_device_data(field_new), &
                               field_device_data(field_old), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_time_smooth: ' // dlesm_error_text())
  end subroutine invoke_time_smooth

  !> The periodic-boundary copies of a field -- its halo(:) list (field_mod.f90:1394-1464), in order --
  !! on the device: what a periodic model's PSy layer does after every kernel that writes the field.
  subroutine invoke_periodic_halos(fld)
    type(r2d_field), intent(inout), target :: fld
    type(c_region) :: cint
    integer(c_int) :: rc
    if (fld%num_halos == 0) return
    call need_device(fld)
    associate (it => fld%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_periodic_halos_apply_f64(field_device_data(fld), int(fld%grid%nx, c_int), int(fld%grid%ny, c_int), &
                                        cint, int(fld%grid%boundary_conditions(1), c_int), &
                                        int(fld%grid%boundary_conditions(2), c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_periodic_halos: ' // dlesm_error_text())
  end subroutine invoke_periodic_halos

  !> invoke_periodic_halos of up to four fields of one grid and internal region in TWO launches (all x copies, then all y
  !! copies) instead of two per field: what follows compute_cu / cv / z / h, or the three kernels of the new time level
  subroutine invoke_periodic_halos_multi(f1, f2, f3, f4)
    type(r2d_field), intent(inout), target :: f1, f2
    type(r2d_field), intent(inout), target, optional :: f3, f4
    type(c_ptr) :: ptrs(4)
    type(c_region) :: cint
    integer(c_int) :: rc, n
    if (f1%num_halos == 0) return
    call need_device(f1);  call need_device(f2)
    ptrs(1) = field_device_data(f1);  ptrs(2) = field_device_data(f2);  n = 2
    if (present(f3)) then
       call need_device(f3);  n = n + 1;  ptrs(n) = field_device_data(f3)
    end if
    if (present(f4)) then
       call need_device(f4);  n = n + 1;  ptrs(n) = field_device_data(f4)
    end if
    associate (it => f1%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_periodic_halos_apply_multi_f64(ptrs, n, int(f1%grid%nx, c_int), int(f1%grid%ny, c_int), cint, &
                                              int(f1%grid%boundary_conditions(1), c_int), &
                                              int(f1%grid%boundary_conditions(2), c_int), c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_periodic_halos_multi: ' // dlesm_error_text())
  end subroutine invoke_periodic_halos_multi

  !> Optional planning call for invoke_shallow_step (like plan_jacobi5): the library times its
  !! launch shapes and cache policies on these fields once -- every trial is the same valid step
  !! into unew, vnew, pnew -- and keeps the fastest for this field geometry.
  subroutine plan_shallow_step(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_autotune_f64(prm, int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                                    int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                                    int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                                    field_device_data(u), field_device_data(v), field_device_data(p), &
                                    field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                                    field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                                    c_null_ptr)
    if (rc /= 0) call gocean_stop('plan_shallow_step: ' // dlesm_error_text())
  end subroutine plan_shallow_step

  !> Distributed shallow-water step: u, v, p must have valid halos; unew, vnew, pnew leave with
  !! theirs, exchanged in ONE grouped RCCL launch that runs behind the interior sweep.
  subroutine invoke_shallow_step_dm(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) then
       call invoke_shallow_step(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
       return
    end if
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_step_dm(halo_plan_for(p%grid%nx, p%grid%ny), prm, &
                               int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                               int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                               int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                               field_device_data(u), field_device_data(v), field_device_data(p), &
                               field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                               field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                               c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_dm: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_dm

  !> The same step for a time loop: returns with the exchange of the new fields in flight; the next
  !! invoke_shallow_step_dm_pipelined on this grid waits for it on the device; halo_join(grid) after the loop.
  subroutine invoke_shallow_step_dm_pipelined(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(c_sw_params), intent(in) :: prm
    type(r2d_field), intent(inout), target :: u, v, p, uold, vold, pold, unew, vnew, pnew
    integer(c_int) :: rc
    if (.not. DIST_MEM_ENABLED) then
       call invoke_shallow_step(prm, u, v, p, uold, vold, pold, unew, vnew, pnew)
       return
    end if
    call need_device(u);  call need_device(v);  call need_device(p)
    call need_device(uold);  call need_device(vold);  call need_device(pold)
    call need_device(unew);  call need_device(vnew);  call need_device(pnew)
    rc = dlesm_shallow_step_dm_pipelined(halo_plan_for(p%grid%nx, p%grid%ny), prm, &
                               int(p%grid%nx, c_int), int(p%grid%ny, c_int), &
                               int(p%internal%xstart, c_int), int(p%internal%xstop, c_int), &
                               int(p%internal%ystart, c_int), int(p%internal%ystop, c_int), &
                               field_device_data(u), field_device_data(v), field_device_data(p), &
                               field_device_data(uold), field_device_data(vold), field_device_data(pold), &
                               field_device_data(unew), field_device_data(vnew), field_device_data(pnew), &
                               c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_shallow_step_dm_pipelined: ' // dlesm_error_text())
  end subroutine invoke_shallow_step_dm_pipelined

  !> halo_exchange(1) of several fields of one grid in a single grouped RCCL launch
  subroutine halo_exchange_multi(f1, f2, f3)
    use parallel_comms_mod, only: halo_plan_for
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(r2d_field), intent(inout), target :: f1, f2
    type(r2d_field), intent(inout), target, optional :: f3
    type(c_ptr) :: ptrs(3)
    integer(c_int) :: rc, n
    if (.not. DIST_MEM_ENABLED) return
    call need_device(f1);  call need_device(f2)
    ptrs(1) = field_device_data(f1);  ptrs(2) = field_device_data(f2);  n = 2
    if (present(f3)) then
       call need_device(f3)
       ptrs(3) = field_device_data(f3);  n = 3
    end if
    rc = dlesm_halo_exchange_multi_f64(halo_plan_for(f1%grid%nx, f1%grid%ny), ptrs, n, DLESM_DIRS_ALL, c_null_ptr)
    if (rc /= 0) call gocean_stop('halo_exchange_multi: ' // dlesm_error_text())
  end subroutine halo_exchange_multi

  !> Device mirrors of the grid properties a kernel may request through its metadata
  !! (GO_GRID_MASK_T, GO_GRID_DX_T, GO_GRID_AREA_T, ... argument_mod): fills the `*_device`
  !! pointers of grid_type (reference grid_mod.f90:104-150) with HBM copies of the host arrays.
  !! Idempotent; arrays keep the field layout (grid%nx x grid%ny, column-major).
  subroutine grid_to_device(grid)
    type(grid_type), intent(inout), target :: grid
    integer(c_size_t) :: nb
    nb = int(grid%nx, c_size_t) * int(grid%ny, c_size_t) * 8_c_size_t
    if (.not. c_associated(grid%tmask_device) .and. allocated(grid%tmask)) then
       if (hipMalloc(grid%tmask_device, nb / 2) /= 0) call gocean_stop('grid_to_device: hipMalloc failed')
       if (hipMemcpy(grid%tmask_device, c_loc(grid%tmask), nb / 2, 1_c_int) /= 0) &
            call gocean_stop('grid_to_device: upload failed')
    end if
    call mirror(grid%dx_t, grid%dx_t_device);  call mirror(grid%dy_t, grid%dy_t_device)
    call mirror(grid%dx_u, grid%dx_u_device);  call mirror(grid%dy_u, grid%dy_u_device)
    call mirror(grid%dx_v, grid%dx_v_device);  call mirror(grid%dy_v, grid%dy_v_device)
    call mirror(grid%dx_f, grid%dx_f_device);  call mirror(grid%dy_f, grid%dy_f_device)
    call mirror(grid%area_t, grid%area_t_device);  call mirror(grid%area_u, grid%area_u_device)
    call mirror(grid%area_v, grid%area_v_device)
    call mirror(grid%gphiu, grid%gphiu_device);  call mirror(grid%gphiv, grid%gphiv_device)
    call mirror(grid%gphif, grid%gphif_device)
    call mirror(grid%xt, grid%xt_device);  call mirror(grid%yt, grid%yt_device)
  contains
    subroutine mirror(host, dev)
      real(go_wp), allocatable, target, intent(in) :: host(:,:)
      type(c_ptr), intent(inout) :: dev
      if (c_associated(dev) .or. .not. allocated(host)) return
      if (hipMalloc(dev, nb) /= 0) call gocean_stop('grid_to_device: hipMalloc failed')
      if (hipMemcpy(dev, c_loc(host), nb, 1_c_int) /= 0) call gocean_stop('grid_to_device: upload failed')
    end subroutine mirror
  end subroutine grid_to_device

  !> The `copy` kernel of infrastructure_mod over all points
  subroutine invoke_copy(out, in)
    type(r2d_field), intent(inout), target :: out, in
    call need_device(in);  call need_device(out)
    call copy_field(in, out)
  end subroutine invoke_copy

  !> Synthetic initial condition: a hash of the GLOBAL cell index on the field's whole region
  subroutine invoke_hash_init(fld, seed, internal_only)
    type(r2d_field), intent(inout), target :: fld
    integer(c_int64_t), intent(in) :: seed
    logical, intent(in), optional :: internal_only    !< fill the internal region only (default: whole)
    integer(c_int) :: rc
    integer :: bx0, bx1, by0, by1
    integer(c_int64_t) :: gx0, gy0
    call need_device(fld)
    gx0 = fld%grid%subdomain%global%xstart - fld%grid%subdomain%internal%xstart + 1
    gy0 = fld%grid%subdomain%global%ystart - fld%grid%subdomain%internal%ystart + 1
    bx0 = fld%whole%xstart;  bx1 = fld%whole%xstop;  by0 = fld%whole%ystart;  by1 = fld%whole%ystop
    if (present(internal_only)) then
       if (internal_only) then
          bx0 = fld%internal%xstart;  bx1 = fld%internal%xstop
          by0 = fld%internal%ystart;  by1 = fld%internal%ystop
       end if
    end if
    rc = dlesm_hash_init_f64(field_device_data(fld), int(fld%grid%nx, c_int), int(fld%grid%ny, c_int), &
                             int(bx0, c_int), int(bx1, c_int), int(by0, c_int), int(by1, c_int), seed, gx0, gy0, &
                             c_null_ptr)
    if (rc /= 0) call gocean_stop('invoke_hash_init: ' // dlesm_error_text())
  end subroutine invoke_hash_init

end module dlesm_psy_mod
