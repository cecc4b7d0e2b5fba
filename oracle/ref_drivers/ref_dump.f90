!> TEST INFRASTRUCTURE (oracle).  Dump driver linked against the REAL reference
!! library (stfc/dl_esm_inf serial build in oracle/_ref) -- written for this
!! repository, it only *calls* the reference's public API and prints what it
!! returns, one "G: key values..." line per fact, so that oracle/make_golden.py
!! can turn the lines into tests/golden/*.json.
!!
!! Usage:
!!   ref_dump.exe decomp  NX NY NDOM
!!   ref_dump.exe bounds  NX NY OFFSET BCX BCY PTYPE     (DL_ESM_ALIGNMENT from env)
!!   ref_dump.exe model   NX NY FILL                     (config-1 plumbing run)
!!   ref_dump.exe gather  NX NY                          (scatter + gather, 1 rank)
program ref_dump
  use kind_params_mod
  use parallel_mod
  use grid_mod
  use field_mod
  use gocean_mod
  use decomposition_mod, only: decomposition_type
  implicit none
  character(len=32) :: cmd, arg
  integer :: nx, ny, ndom, offset, bcx, bcy, ptype
  real(go_wp) :: fill

  call get_command_argument(1, cmd)
  call get_command_argument(2, arg); read(arg, *) nx
  call get_command_argument(3, arg); read(arg, *) ny

  call gocean_initialise()

  select case (trim(cmd))
  case ('decomp')
     call get_command_argument(4, arg); read(arg, *) ndom
     call dump_decomp(nx, ny, ndom)
  case ('bounds')
     call get_command_argument(4, arg); read(arg, *) offset
     call get_command_argument(5, arg); read(arg, *) bcx
     call get_command_argument(6, arg); read(arg, *) bcy
     call get_command_argument(7, arg); read(arg, *) ptype
     call dump_bounds(nx, ny, offset, bcx, bcy, ptype)
  case ('model')
     call get_command_argument(4, arg); read(arg, *) fill
     call dump_model(nx, ny, fill)
  case ('gather')
     call dump_gather(nx, ny)
  case ('tmask')
     call dump_tmask(nx, ny)
  case default
     stop 'ref_dump: unknown command'
  end select

  call gocean_finalise()

contains

  subroutine dump_decomp(nx, ny, ndom)
    integer, intent(in) :: nx, ny, ndom
    type(decomposition_type) :: d
    integer :: i
    d = go_decompose(nx, ny, ndomains=ndom)
    write(*, '("G: decomp ",7(I0,1x))') d%global_nx, d%global_ny, d%nx, d%ny, &
         d%ndomains, d%max_width, d%max_height
    do i = 1, d%ndomains
       write(*, '("G: sub ",I0,12(1x,I0))') i, &
            d%subdomains(i)%global%xstart, d%subdomains(i)%global%xstop, &
            d%subdomains(i)%global%ystart, d%subdomains(i)%global%ystop, &
            d%subdomains(i)%global%nx, d%subdomains(i)%global%ny, &
            d%subdomains(i)%internal%xstart, d%subdomains(i)%internal%xstop, &
            d%subdomains(i)%internal%ystart, d%subdomains(i)%internal%ystop, &
            d%subdomains(i)%internal%nx, d%subdomains(i)%internal%ny
    end do
  end subroutine dump_decomp

  subroutine dump_bounds(nx, ny, offset, bcx, bcy, ptype)
    integer, intent(in) :: nx, ny, offset, bcx, bcy, ptype
    type(grid_type), target :: g
    type(r2d_field) :: f
    g = grid_type(GO_ARAKAWA_C, (/bcx, bcy, GO_BC_NONE/), offset)
    call g%decompose(nx, ny)
    call grid_init(g, 1.0_go_wp, 1.0_go_wp)
    write(*, '("G: grid ",4(I0,1x))') g%nx, g%ny, g%global_nx, g%global_ny
    f = r2d_field(g, ptype)
    write(*, '("G: field ",I0,13(1x,I0))') f%defined_on, &
         f%internal%xstart, f%internal%xstop, f%internal%ystart, f%internal%ystop, &
         f%internal%nx, f%internal%ny, &
         f%whole%xstart, f%whole%xstop, f%whole%ystart, f%whole%ystop, &
         f%whole%nx, f%whole%ny, f%num_halos
    write(*, '("G: shape ",2(I0,1x))') size(f%data, 1), size(f%data, 2)
    if (allocated(f%halo)) call dump_halos(f)
  end subroutine dump_bounds

  subroutine dump_halos(f)
    type(r2d_field), intent(in) :: f
    integer :: i
    do i = 1, f%num_halos
       write(*, '("G: halo ",I0,8(1x,I0))') i, &
            f%halo(i)%source%xstart, f%halo(i)%source%xstop, &
            f%halo(i)%source%ystart, f%halo(i)%source%ystop, &
            f%halo(i)%dest%xstart, f%halo(i)%dest%xstop, &
            f%halo(i)%dest%ystart, f%halo(i)%dest%ystop
    end do
  end subroutine dump_halos

  !> BASELINE.json configs[0]: the example model scaled to NX x NY, one T field.
  subroutine dump_model(nx, ny, fill)
    integer, intent(in) :: nx, ny
    real(go_wp), intent(in) :: fill
    type(grid_type), target :: g
    type(r2d_field) :: t
    integer, allocatable :: tmask(:,:)
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), &
                  GO_OFFSET_NE)
    call g%decompose(nx, ny)
    allocate(tmask(g%subdomain%global%nx, g%subdomain%global%ny))
    tmask(:,:) = 1
    call grid_init(g, 1.0_go_wp, 1.0_go_wp, tmask)
    t = r2d_field(g, GO_T_POINTS)
    t%data(:,:) = fill
    call t%halo_exchange(1)
    write(*, '("G: grid ",4(I0,1x))') g%nx, g%ny, g%global_nx, g%global_ny
    write(*, '("G: internal ",4(I0,1x))') t%internal%xstart, t%internal%xstop, &
         t%internal%ystart, t%internal%ystop
    write(*, '("G: checksum ",ES24.16E3)') field_checksum(t)
    write(*, '("G: xt ",3(ES24.16E3,1x))') g%xt(1,1), g%xt(2,1), g%xt(g%nx,1)
    write(*, '("G: yt ",3(ES24.16E3,1x))') g%yt(1,1), g%yt(1,2), g%yt(1,g%ny)
  end subroutine dump_model

  !> grid_init with a patterned T mask (values -1, 0, 1): the grid's tmask after the copy-in and
  !! the boundary fill (grid_mod.f90:394-432), row by row.
  subroutine dump_tmask(nx, ny)
    integer, intent(in) :: nx, ny
    type(grid_type), target :: g
    integer, allocatable :: tmask(:,:)
    integer :: i, j
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), &
                  GO_OFFSET_NE)
    call g%decompose(nx, ny)
    allocate(tmask(g%subdomain%global%nx, g%subdomain%global%ny))
    do j = 1, size(tmask, 2)
       do i = 1, size(tmask, 1)
          tmask(i, j) = mod(7*i + 13*j, 3) - 1
       end do
    end do
    call grid_init(g, 1.0_go_wp, 1.0_go_wp, tmask)
    write(*, '("G: grid ",4(I0,1x))') g%nx, g%ny, g%global_nx, g%global_ny
    do j = 1, g%ny
       write(*, '("G: tmaskrow ",I0,*(1x,I0))') j, (g%tmask(i, j), i = 1, g%nx)
    end do
  end subroutine dump_tmask

  !> init_global_data scatter followed by gather_inner_data on one rank.
  subroutine dump_gather(nx, ny)
    integer, intent(in) :: nx, ny
    type(grid_type), target :: g
    type(r2d_field) :: t
    real(go_wp), allocatable :: glob(:,:), back(:,:)
    integer :: i, j, nbad
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), &
                  GO_OFFSET_NE)
    call g%decompose(nx, ny)
    call grid_init(g, 1.0_go_wp, 1.0_go_wp)
    allocate(glob(nx, ny))
    do j = 1, ny
       do i = 1, nx
          glob(i, j) = real((i - 1) + (j - 1)*nx, go_wp)
       end do
    end do
    t = r2d_field(g, GO_T_POINTS, init_global_data=glob)
    write(*, '("G: corner ",4(ES24.16E3,1x))') t%data(1,1), t%data(2,2), &
         t%data(nx+1, ny+1), t%data(nx+2, ny+2)
    write(*, '("G: checksum ",ES24.16E3)') field_checksum(t)
    call t%gather_inner_data(back)
    nbad = count(back /= glob)
    write(*, '("G: gather_shape ",2(I0,1x))') size(back, 1), size(back, 2)
    write(*, '("G: gather_mismatch ",I0)') nbad
  end subroutine dump_gather

end program ref_dump
