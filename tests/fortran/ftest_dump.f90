!> Dump driver for THIS library's Fortran API layer (lib_fd_hip.a): the same commands and the
!! same "G: key values" output as oracle/ref_drivers/ref_dump.f90 prints for the reference, so
!! that tests/test_fortran_layer.py can compare the two line by line.  Needs no GPU.
!!
!!   ftest_dump.exe decomp NX NY NDOM
!!   ftest_dump.exe bounds NX NY OFFSET BCX BCY PTYPE
!!   ftest_dump.exe model  NX NY FILL
!!   ftest_dump.exe gather NX NY
!!   ftest_dump.exe comms  NX NY          (RANK / WORLD_SIZE from the environment, DLESM_DRY_COMMS=1)
program ftest_dump
  use kind_params_mod
  use parallel_mod
  use grid_mod
  use field_mod
  use gocean_mod
  use decomposition_mod, only: decomposition_type
  use parallel_comms_mod
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
  case ('comms')
     ndom = 1                                   ! optional 4th argument: halo width of the decomposition
     if (command_argument_count() >= 4) then
        call get_command_argument(4, arg); read(arg, *) ndom
     end if
     call dump_comms(nx, ny, ndom)
  case default
     stop 'ftest_dump: unknown command'
  end select
  call gocean_finalise()

contains

  subroutine dump_decomp(nx, ny, ndom)
    integer, intent(in) :: nx, ny, ndom
    type(decomposition_type) :: d
    integer :: i
    d = go_decompose(nx, ny, ndomains=ndom)
    write(*, '("G: decomp ",7(I0,1x))') d%global_nx, d%global_ny, d%nx, d%ny, d%ndomains, &
         d%max_width, d%max_height
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
    integer :: i
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
    if (allocated(f%halo)) then
       do i = 1, f%num_halos
          write(*, '("G: halo ",I0,8(1x,I0))') i, &
               f%halo(i)%source%xstart, f%halo(i)%source%xstop, &
               f%halo(i)%source%ystart, f%halo(i)%source%ystop, &
               f%halo(i)%dest%xstart, f%halo(i)%dest%xstop, &
               f%halo(i)%dest%ystart, f%halo(i)%dest%ystop
       end do
    end if
  end subroutine dump_bounds

  subroutine dump_model(nx, ny, fill)
    integer, intent(in) :: nx, ny
    real(go_wp), intent(in) :: fill
    type(grid_type), target :: g
    type(r2d_field) :: t
    integer, allocatable :: tmask(:,:)
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), GO_OFFSET_NE)
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
  !! the boundary fill, row by row -- same "G:" lines as oracle/ref_drivers/ref_dump.f90 prints
  !! for the real reference.
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

  subroutine dump_gather(nx, ny)
    integer, intent(in) :: nx, ny
    type(grid_type), target :: g
    type(r2d_field) :: t
    real(go_wp), allocatable :: glob(:,:), back(:,:)
    integer :: i, j
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), GO_OFFSET_NE)
    call g%decompose(nx, ny)
    call grid_init(g, 1.0_go_wp, 1.0_go_wp)
    allocate(glob(nx, ny))
    do j = 1, ny
       do i = 1, nx
          glob(i, j) = real((i - 1) + (j - 1)*nx, go_wp)
       end do
    end do
    t = r2d_field(g, GO_T_POINTS, init_global_data=glob)
    write(*, '("G: corner ",4(ES24.16E3,1x))') t%data(1,1), t%data(2,2), t%data(nx+1, ny+1), &
         t%data(nx+2, ny+2)
    write(*, '("G: checksum ",ES24.16E3)') field_checksum(t)
    call t%gather_inner_data(back)
    write(*, '("G: gather_shape ",2(I0,1x))') size(back, 1), size(back, 2)
    write(*, '("G: gather_mismatch ",I0)') count(back /= glob)
  end subroutine dump_gather

  !> this rank's message tables after grid_init (dry communicator)
  subroutine dump_comms(nx, ny, hw)
    integer, intent(in) :: nx, ny, hw
    type(grid_type), target :: g
    integer :: k
    g = grid_type(GO_ARAKAWA_C, (/GO_BC_EXTERNAL, GO_BC_EXTERNAL, GO_BC_NONE/), GO_OFFSET_NE)
    call g%decompose(nx, ny, halo_width=hw)
    call grid_init(g, 1.0_go_wp, 1.0_go_wp)
    write(*, '("G: rank ",4(I0,1x))') get_rank(), get_num_ranks(), g%nx, g%ny
    write(*, '("G: counts ",2(I0,1x))') nsend, nrecv
    do k = 1, nsend
       write(*, '("G: send ",8(I0,1x))') dirsend(k), destination(k), isrcsend(k), jsrcsend(k), &
            idessend(k), jdessend(k), nxsend(k), nysend(k)
    end do
    do k = 1, nrecv
       write(*, '("G: recv ",6(I0,1x))') dirrecv(k), source(k), idesrecv(k), jdesrecv(k), &
            nxrecv(k), nyrecv(k)
    end do
    write(*, '("G: bounds ",4(I0,1x))') ielb, ieub, jelb, jeub
  end subroutine dump_comms

end program ftest_dump
