!> A GOcean-style application on the MI355X library: algorithm layer in plain Fortran with
!! the dl_esm_inf API (grid_type, r2d_field, halo_exchange, field_checksum), PSy layer =
!! the HIP launch wrappers of dlesm_psy_mod.  5-point Jacobi ping-pong on a T field.
!!
!!   jacobi_app.exe [N] [NSTEPS] [FUSE] [PLAN] [PEER]  (default 4096 100 1 1 0; DL_ESM_ALIGNMENT honoured; PLAN = 0 skips
!!                        the optional planning call; PEER = 1 connects the ranks' mailboxes, the distributed steps then
!!                        exchange with stores over xGMI instead of an RCCL group; ten untimed warm-up steps precede the loop)
!! One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT in the environment (e.g.
!!   for r in 0 1; do RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r MASTER_PORT=29400 ./jacobi_app.exe & done).
!! With more than one rank the global domain is (N*P) x (N*Q) so that every rank owns N x N.
!> The kernel as a GOcean kernel module: metadata type (what PSyclone reads at code-generation
!! time: access modes, grid-point types, the 5-point stencil, iteration space) + the pointwise
!! code.  On MI355X the PSy layer launches the HIP implementation of exactly this kernel
!! (invoke_jacobi5 / invoke_jacobi5_dm) instead of looping over jacobi5_code.
module jacobi5_mod
  use kind_params_mod
  use kernel_mod
  use argument_mod
  use grid_mod
  implicit none
  type, extends(kernel_type) :: jacobi5
     type(go_arg), dimension(2) :: meta_args =                      &
          (/ go_arg(GO_WRITE, GO_CT, GO_POINTWISE),                  & ! out
             go_arg(GO_READ,  GO_CT, go_stencil(010, 101, 010)) /)     ! in: W, E, S, N neighbours
     integer :: ITERATES_OVER = GO_INTERNAL_PTS
     integer :: index_offset = GO_OFFSET_ANY
   contains
     procedure, nopass :: code => jacobi5_code
  end type jacobi5
contains
  subroutine jacobi5_code(ji, jj, out, in)
    integer, intent(in) :: ji, jj
    real(go_wp), dimension(:,:), intent(out) :: out
    real(go_wp), dimension(:,:), intent(in) :: in
    out(ji, jj) = 0.25_go_wp * ((in(ji-1, jj) + in(ji+1, jj)) + (in(ji, jj-1) + in(ji, jj+1)))
  end subroutine jacobi5_code
end module jacobi5_mod

program jacobi_app
  use iso_c_binding
  use jacobi5_mod, only: jacobi5
  use kind_params_mod
  use parallel_mod
  use grid_mod
  use field_mod
  use gocean_mod
  use dlesm_psy_mod
  implicit none
  character(len=32) :: arg
  integer :: n, nsteps, i, p, q, nr, fuse, ncalls, plan, i0, peer
  integer(8) :: t0, t1, rate
  type(grid_type), target :: model_grid
  type(r2d_field), target :: a, b
  real(go_wp) :: cs, secs

  ! jacobi_app [n [nsteps [fuse]]]: tile n x n per rank, nsteps time steps, `fuse` (1..8) of them
  ! advanced per sweep (temporal blocking; one depth-`fuse` halo exchange per sweep)
  n = 4096;  nsteps = 100;  fuse = 1
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg);  read(arg, *) n
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg);  read(arg, *) nsteps
  end if
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg);  read(arg, *) fuse
  end if
  plan = 1
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg);  read(arg, *) plan
  end if
  peer = 0
  if (command_argument_count() >= 5) then
     call get_command_argument(5, arg);  read(arg, *) peer
  end if
  if (fuse < 1 .or. fuse > 8) stop 'jacobi_app: fuse must be 1..8'
  nsteps = (nsteps / fuse) * fuse
  ncalls = nsteps / fuse

  call gocean_initialise()
  nr = get_num_ranks()
  p = int(sqrt(real(nr)))
  do while (mod(nr, p) /= 0)
     p = p - 1
  end do
  q = nr / p

  model_grid = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), GO_OFFSET_NE)
  call model_grid%decompose(n * p, n * q, halo_width=fuse)
  call grid_init(model_grid, 1.0_go_wp, 1.0_go_wp)
  a = r2d_field(model_grid, GO_T_POINTS)
  b = r2d_field(model_grid, GO_T_POINTS)

  ! ---- PSy layer --------------------------------------------------------------------
  call invoke_hash_init(a, 20261004_c_int64_t)   ! includes the fixed boundary ring
  call invoke_copy(b, a)
  call a%halo_exchange(1)
  call model_write_log("('initial checksum = ',E24.16)", field_checksum(a))

  if (fuse == 1 .and. peer /= 0) call halo_connect_peers(model_grid)     ! collective; no-op on one rank
  if (fuse == 1 .and. plan /= 0) then
     call plan_jacobi5(b, a)            ! optional: the library times its launch shapes once
     call invoke_copy(b, a)
  end if
  ! ten untimed warm-up steps (an even number: a and b keep their roles), then the timed loop
  do i0 = 0, 1
  if (i0 == 1) then
     call halo_join(model_grid)
     call device_sync()
     call system_clock(t0, rate)
  end if
  do i = 1, merge(10 / fuse + mod(10 / fuse, 2), ncalls, i0 == 0)
     if (fuse == 1) then
        if (mod(i, 2) == 1) then
           call invoke_jacobi5_dm_pipelined(b, a)   ! exchange of the result hidden behind the interior
        else
           call invoke_jacobi5_dm_pipelined(a, b)
        end if
     else
        if (mod(i, 2) == 1) then
           call invoke_jacobi5_multi(b, a, fuse)
        else
           call invoke_jacobi5_multi(a, b, fuse)
        end if
     end if
  end do
  end do
  call halo_join(model_grid)        ! the one join of the time loop (no-op on one rank)
  call device_sync()
  call system_clock(t1)
  secs = real(t1 - t0, go_wp) / real(rate, go_wp)

  if (mod(ncalls, 2) == 1) then
     cs = field_checksum(b)
  else
     cs = field_checksum(a)
  end if
  call model_write_log("('final checksum   = ',E24.16)", cs)
  call model_write_log("('Mcells/s (all ranks) = ',F14.1)", &
       real(n, go_wp) * real(n, go_wp) * real(nr, go_wp) * real(nsteps, go_wp) / secs / 1.0e6_go_wp)
  call free_field(a);  call free_field(b)
  call gocean_finalise()
end program jacobi_app
