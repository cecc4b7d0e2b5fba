!> SURVEY section 8(d), last clause: "the real /root/reference build is additionally timed HERE (not on the GPU box) for
!! the ... init paths".  One program, written against the API both libraries share (grid_type / decompose / grid_init /
!! r2d_field, the call sequence of the reference's example/model.f90:54-85), compiled TWICE by scripts/init_path_timing.sh:
!! against oracle/_ref (the reference's own serial sources, grid_mod.f90:330-570 + field_mod.f90:242-390) and against this
!! repository's Fortran layer (dl_esm_inf_amd/fortran).  Prints wall-clock seconds of grid_init and of the four r2d_field
!! constructors, plus the checksum of a field set to 1.0 (must be N*N on both sides).
program init_path_timing
  use kind_params_mod
  use grid_mod
  use field_mod
  use gocean_mod
  implicit none
  integer :: n, ierr, reps, r
  character(len=32) :: arg
  type(grid_type), target :: g
  type(r2d_field) :: fu, fv, ft, ff
  integer, allocatable :: tmask(:,:)
  integer(kind=8) :: c0, c1, c2, c3, rate
  real(go_wp) :: cs

  n = 4096
  reps = 1
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg)
     read(arg, *) n
  end if
  call gocean_initialise()
  call system_clock(c0, rate)
  g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), GO_OFFSET_NE)
  call g%decompose(n, n)
  allocate(tmask(g%subdomain%global%nx, g%subdomain%global%ny), stat=ierr)
  if (ierr /= 0) call gocean_stop('tmask')
  tmask(:,:) = 1
  call system_clock(c1)
  call grid_init(g, 1.0_go_wp, 1.0_go_wp, tmask)
  call system_clock(c2)
  fu = r2d_field(g, GO_U_POINTS)
  fv = r2d_field(g, GO_V_POINTS)
  ft = r2d_field(g, GO_T_POINTS)
  ff = r2d_field(g, GO_F_POINTS)
  call system_clock(c3)
  ft%data(ft%internal%xstart:ft%internal%xstop, ft%internal%ystart:ft%internal%ystop) = 1.0_go_wp
  cs = field_checksum(ft)
  write(*, '("N=",I6," nx=",I6," ny=",I6," decompose+tmask_s=",F9.4," grid_init_s=",F9.4," four_fields_s=",F9.4," checksum=",ES23.15)') &
       n, g%nx, g%ny, real(c1 - c0, 8) / real(rate, 8), real(c2 - c1, 8) / real(rate, 8), real(c3 - c2, 8) / real(rate, 8), cs
  call gocean_finalise()
end program init_path_timing
