!> A GOcean-style shallow-water model on the dl_esm_inf API, in the configuration of the public
!! GOcean `shallow` benchmark: Arakawa C grid, SW offset, periodic in x and y (the only periodic
!! configuration the reference supports, serially: field_mod.f90:675-751, grid_mod.f90:437-442).
!! Algorithm layer = the usual sequence -- build grid and fields, initialise, time loop of
!! [u/v/h update; periodic boundary copies of the new fields; leapfrog rotation], checksums --
!! PSy layer = the launch wrappers of dlesm_psy_mod instead of loop nests over compute_*_code, in one of three
!! forms (MODE, default 0):
!!   0  fused     one launch per time step: invoke_shallow_step_sw_periodic (step + periodic images)
!!   1  kernels   what an UNMODIFIED generated PSy layer does: one launch per GOcean kernel -- compute_cu, cv, z,
!!                h, the periodic copies of the four intermediates, compute_unew, vnew, pnew, the periodic copies of
!!                the new level -- seven kernel launches per step, every intermediate through HBM
!!   2  kernels + time_smooth: mode 1 followed by the Asselin filter of the old level (alpha = 0.001), as the
!!                benchmark's time loop has it; uold <- smoothed u, u <- unew by rotation
!!   3  fused + time_smooth: ONE launch per time step (invoke_shallow_step_sw_smooth_periodic: update, Asselin filter of the
!!                old level, periodic images of both levels)
!!   4  TWO whole time steps per launch (invoke_shallow_step_sw_smooth_x2_periodic: update, filter and periodic images, twice;
!!                48 instead of 96 B/cell/step); the loop ping-pongs between two sextets of fields, an odd last step is one
!!                mode-3 launch
!! Modes 0 and 1 print the same bits; so do modes 2, 3 and 4.
!!     shallow_app.exe N NSTEPS [MODE]
program shallow_app
  use iso_c_binding
  use kind_params_mod
  use parallel_mod
  use grid_mod
  use field_mod
  use gocean_mod
  use dlesm_psy_mod
  implicit none
  character(len=32) :: arg
  integer :: n, nsteps, step, k, rate, t0, t1, mode
  type(grid_type), target :: model_grid
  type(r2d_field), target :: f(12)         ! u v p | uold vold pold | unew vnew pnew | (mode 4) the second old level
  type(r2d_field), target :: cu, cv, z, h  ! the intermediates of the per-kernel forms
  real(go_wp), parameter :: dt = 90.0_go_wp, alpha = 0.001_go_wp
  real(go_wp) :: tdt
  integer :: ptype(12), cur(3), old(3), new(3), tmp(3), od2(3)
  real(go_wp), pointer :: d(:,:)
  real(go_wp) :: secs

  n = 256;  nsteps = 10
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg, *) nsteps
  end if
  mode = 0
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg); read(arg, *) mode
  end if
  tdt = dt + dt
  call gocean_initialise()
  model_grid = grid_type(GO_ARAKAWA_C, (/GO_BC_PERIODIC, GO_BC_PERIODIC, GO_BC_NONE/), GO_OFFSET_SW)
  call model_grid%decompose(n, n)
  call grid_init(model_grid, 1.0e5_go_wp, 1.0e5_go_wp)
  ptype = (/ GO_U_POINTS, GO_V_POINTS, GO_T_POINTS, GO_U_POINTS, GO_V_POINTS, GO_T_POINTS, &
             GO_U_POINTS, GO_V_POINTS, GO_T_POINTS, GO_U_POINTS, GO_V_POINTS, GO_T_POINTS /)
  do k = 1, merge(12, 9, mode == 4)
     f(k) = r2d_field(model_grid, ptype(k))
  end do
  ! initial state: counter hash on the internal region (u, v in [-0.5,0.5), p in [1,2)), periodic halos,
  ! old and new levels start as copies
  do k = 1, 3
     call invoke_hash_init(f(k), int(100 + k, c_int64_t), internal_only=.true.)
     d => f(k)%get_data()
     if (k == 3) then
        d = d + 1.0_go_wp
     else
        d = d - 0.5_go_wp
     end if
     call f(k)%write_to_device()
     call invoke_periodic_halos(f(k))
     call invoke_copy(f(k + 3), f(k))
     call invoke_copy(f(k + 6), f(k))
     if (mode == 4) call invoke_copy(f(k + 9), f(k))
  end do
  cur = (/1, 2, 3/);  old = (/4, 5, 6/);  new = (/7, 8, 9/);  od2 = (/10, 11, 12/)
  if (mode == 1 .or. mode == 2) then
     cu = r2d_field(model_grid, GO_U_POINTS);  cv = r2d_field(model_grid, GO_V_POINTS)
     z = r2d_field(model_grid, GO_F_POINTS);   h = r2d_field(model_grid, GO_T_POINTS)
     ! place the four intermediates in HBM now (the PSy wrappers would otherwise allocate and upload them at their first
     ! use, inside the timed loop: 4 x 537 MB over PCIe at 8192^2)
     call invoke_copy(cu, f(1));  call invoke_copy(cv, f(1));  call invoke_copy(z, f(1));  call invoke_copy(h, f(1))
  else
     ! planning call (once, outside the time loop): the new level receives one valid step
     call plan_shallow_step_sw(shallow_params(model_grid%dx, model_grid%dy, dt), &
                               f(cur(1)), f(cur(2)), f(cur(3)), f(old(1)), f(old(2)), f(old(3)), &
                               f(new(1)), f(new(2)), f(new(3)))
  end if
  call device_sync()
  call system_clock(t0, rate)
  do step = 1, nsteps
     if (mode == 4) then
        if (mod(step, 2) == 0) cycle                       ! (the launch of the odd step before has done this one too)
        if (step < nsteps) then
           call invoke_shallow_step_sw_smooth_x2_periodic(shallow_params(model_grid%dx, model_grid%dy, dt), alpha, &
                                                          f(cur(1)), f(cur(2)), f(cur(3)), f(old(1)), f(old(2)), f(old(3)), &
                                                          f(new(1)), f(new(2)), f(new(3)), f(od2(1)), f(od2(2)), f(od2(3)))
           tmp = cur;  cur = new;  new = tmp               ! level n+2 is current, the filtered level n+1 the old one:
           tmp = old;  old = od2;  od2 = tmp               ! ping-pong between the two sextets
        else                                               ! an odd number of steps: the last one on its own
           call invoke_shallow_step_sw_smooth_periodic(shallow_params(model_grid%dx, model_grid%dy, dt), alpha, &
                                                       f(cur(1)), f(cur(2)), f(cur(3)), f(old(1)), f(old(2)), f(old(3)), &
                                                       f(new(1)), f(new(2)), f(new(3)))
           tmp = cur;  cur = new;  new = tmp
        end if
        cycle
     end if
     if (mode == 0) then
        call invoke_shallow_step_sw_periodic(shallow_params(model_grid%dx, model_grid%dy, dt), &
                                             f(cur(1)), f(cur(2)), f(cur(3)), f(old(1)), f(old(2)), f(old(3)), &
                                             f(new(1)), f(new(2)), f(new(3)))
     else if (mode == 3) then
        call invoke_shallow_step_sw_smooth_periodic(shallow_params(model_grid%dx, model_grid%dy, dt), alpha, &
                                                    f(cur(1)), f(cur(2)), f(cur(3)), f(old(1)), f(old(2)), f(old(3)), &
                                                    f(new(1)), f(new(2)), f(new(3)))
     else
        call invoke_compute_cu(cu, f(cur(3)), f(cur(1)))
        call invoke_compute_cv(cv, f(cur(3)), f(cur(2)))
        call invoke_compute_z(z, f(cur(3)), f(cur(1)), f(cur(2)))
        call invoke_compute_h(h, f(cur(3)), f(cur(1)), f(cur(2)))
        call invoke_periodic_halos_multi(cu, cv, z, h)     ! the periodic copies of the four intermediates: two launches
        call invoke_compute_unew(f(new(1)), f(old(1)), z, cv, h, tdt)
        call invoke_compute_vnew(f(new(2)), f(old(2)), z, cu, h, tdt)
        call invoke_compute_pnew(f(new(3)), f(old(3)), cu, cv, tdt)
        call invoke_periodic_halos_multi(f(new(1)), f(new(2)), f(new(3)))
     end if
     if (mode == 2) then
        do k = 1, 3
           call invoke_time_smooth(f(cur(k)), f(new(k)), f(old(k)), alpha)
        end do
        call invoke_periodic_halos_multi(f(old(1)), f(old(2)), f(old(3)))
     end if
     if (mode >= 2) then
        tmp = cur;  cur = new;  new = tmp                  ! u <- unew; uold holds the smoothed u already
     else
        tmp = old;  old = cur;  cur = new;  new = tmp      ! leapfrog rotation
     end if
  end do
  call device_sync()
  call system_clock(t1)
  secs = real(t1 - t0, go_wp) / real(rate, go_wp)
  write(*, '("G: shape ",3(I0,1x))') model_grid%nx, model_grid%ny, nsteps
  do k = 1, 3
     d => f(cur(k))%get_data()
     write(*, '("G: cs ",I0,1x,3(ES24.16E3,1x))') k, field_checksum(f(cur(k))), d(2, 2), d(n + 1, n + 1)
  end do
  write(*, '("shallow_app: ",I0,"^2, ",I0," steps, ",F10.1," Mcells/s")') n, nsteps, &
       real(n, go_wp) * real(n, go_wp) * nsteps / secs / 1.0e6_go_wp
  call gocean_finalise()
end program shallow_app
