!> Fields on the finite-difference grid: the r2d_field object of the dl_esm_inf API with
!! device-resident data on MI355X.
!!
!! Public names, the components of field_type / r2d_field and the four device-callback
!! interfaces are the reference's (finite_difference/src/field_mod.f90:47-194) so that
!! generated code compiles unchanged.  What differs is behind them:
!!   * iteration bounds come from the C-ABI library (dlesm_field_bounds);
!!   * `device_ptr` points at a dlesm_field descriptor and the C-flavour callbacks are the
!!     library's own (dlesm_read_from_device / dlesm_write_to_device);
!!   * halo_exchange of a device-resident field never leaves the device (RCCL send/recv of
!!     device strips); a host-resident field is exchanged through a device scratch copy;
!!   * field_checksum of a device-resident field is reduced on the device.
module field_mod
  use iso_c_binding
  use kind_params_mod
  use region_mod
  use halo_mod
  use grid_mod
  use gocean_mod, only: gocean_stop
  use tile_mod
  use parallel_mod, only: decomposition_type
  use dlesm_hip_mod
  implicit none
  private

  ! grid-point types of the Arakawa C grid (field_mod.f90:47-52)
  integer, public, parameter :: GO_U_POINTS = 0, GO_V_POINTS = 1, GO_T_POINTS = 2, &
                                GO_F_POINTS = 3, GO_ALL_POINTS = 4

  ! The user-replaceable device transfer hooks (field_mod.f90:65-105): from/to, the 1-based
  ! origin and extent of the patch, and whether to wait for completion.
  abstract interface
     subroutine read_from_device_c_interface(from, to, startx, starty, nx, ny, blocking)
       use iso_c_binding, only: c_ptr, c_int, c_bool
       type(c_ptr), intent(in), value :: from
       type(c_ptr), intent(in), value :: to
       integer(c_int), intent(in), value :: startx, starty, nx, ny
       logical(c_bool), intent(in), value :: blocking
     end subroutine read_from_device_c_interface
     subroutine read_from_device_f_interface(from, to, startx, starty, nx, ny, blocking)
       use iso_c_binding, only: c_ptr
       use kind_params_mod, only: go_wp
       type(c_ptr), intent(in) :: from
       real(go_wp), dimension(:,:), target, intent(inout) :: to
       integer, intent(in) :: startx, starty, nx, ny
       logical, intent(in) :: blocking
     end subroutine read_from_device_f_interface
     subroutine write_to_device_c_interface(from, to, startx, starty, nx, ny, blocking)
       use iso_c_binding, only: c_ptr, c_int, c_bool
       type(c_ptr), intent(in), value :: from
       type(c_ptr), intent(in), value :: to
       integer(c_int), intent(in), value :: startx, starty, nx, ny
       logical(c_bool), intent(in), value :: blocking
     end subroutine write_to_device_c_interface
     subroutine write_to_device_f_interface(from, to, startx, starty, nx, ny, blocking)
       use iso_c_binding, only: c_ptr
       use kind_params_mod, only: go_wp
       real(go_wp), dimension(:,:), target, intent(in) :: from
       type(c_ptr), intent(in) :: to
       integer, intent(in) :: startx, starty, nx, ny
       logical, intent(in) :: blocking
     end subroutine write_to_device_f_interface
  end interface

  type, public :: field_type
     integer :: defined_on
     type(grid_type), pointer :: grid
     type(region_type) :: internal
     type(region_type) :: whole
     integer :: num_halos
     type(halo_type), dimension(:), allocatable :: halo
     logical :: data_on_device
     procedure(read_from_device_c_interface), pointer, nopass :: read_from_device_c
     procedure(read_from_device_f_interface), pointer, nopass :: read_from_device_f
     procedure(write_to_device_c_interface), pointer, nopass :: write_to_device_c
     procedure(write_to_device_f_interface), pointer, nopass :: write_to_device_f
  end type field_type

  type, public, extends(field_type) :: r2d_field
     integer :: ntiles
     type(tile_type), dimension(:), allocatable :: tile
     real(go_wp), dimension(:,:), allocatable :: data
     !> dlesm_field descriptor of the HBM copy (c_null_ptr until first needed)
     type(c_ptr) :: device_ptr
   contains
     procedure, pass :: set_data
     procedure, pass :: get_data
     procedure, pass :: read_halo_from_device
     procedure, pass :: write_halo_to_device
     procedure, pass :: read_from_device
     procedure, pass :: write_to_device
     procedure, public :: halo_exchange
     procedure, public :: gather_inner_data
  end type r2d_field

  interface copy_field
     module procedure copy_2dfield_array, copy_2dfield_array_patch, copy_2dfield, copy_2dfield_patch
  end interface copy_field
  interface r2d_field
     module procedure r2d_field_constructor
  end interface r2d_field
  interface field_checksum
     module procedure fld_checksum, array_checksum
  end interface field_checksum
  interface free_field
     module procedure r2d_free_field
  end interface

  public copy_field, set_field, field_checksum, free_field

  integer, public, parameter :: NBOUNDARY = 1
  logical, public, parameter :: TILED_FIELDS = .FALSE.

  ! additions of this implementation (used by PSy layers that launch HIP kernels)
  public :: field_to_device, field_device_data, field_on_dlesm_device
  public :: dlesm_read_cb, dlesm_write_cb

  type(decomposition_type), save :: tiling
  logical, save :: tiling_initialised = .false.

contains

  function r2d_field_constructor(grid, grid_points, do_tile, init_global_data) result(self)
    use parallel_mod, only: go_decompose, on_master
    type(grid_type), intent(in), target :: grid
    integer, intent(in) :: grid_points
    logical, intent(in), optional :: do_tile
    real(go_wp), dimension(:,:), intent(in), optional :: init_global_data
    type(r2d_field), target :: self
    character(len=8) :: fld_type
    integer :: ierr, itile, ntilex, ntiley, dx, dy, ji, jj
    logical :: tile_it

    self%grid => grid
    self%data_on_device = .false.
    self%device_ptr = c_null_ptr
    nullify(self%read_from_device_c, self%read_from_device_f)
    nullify(self%write_to_device_c, self%write_to_device_f)

    call set_field_bounds(self, fld_type, grid_points)

    tile_it = .false.
    if (present(do_tile)) tile_it = do_tile
    self%ntiles = 0
    if (tile_it) then
       ! one tile per thread; this library is built without OpenMP (as the reference in
       ! practice, field_mod.f90:302-303), so a tiled field has exactly one tile
       if (.not. get_grid_dims(ntilex, ntiley)) then
          ntilex = 1;  ntiley = 1
       end if
       if (.not. tiling_initialised) then
          tiling = go_decompose(self%internal%nx, self%internal%ny, 1, ntilex, ntiley)
          tiling_initialised = .true.
       end if
       self%ntiles = 1
       allocate(self%tile(self%ntiles), stat=ierr)
       if (ierr /= 0) call gocean_stop('r2d constructor failed to allocate tiling structures')
       do itile = 1, self%ntiles
          self%tile(itile)%whole = tiling%subdomains(itile)%global
          self%tile(itile)%internal = tiling%subdomains(itile)%internal
       end do
    end if

    if (on_master()) then
       write(*, "('Allocating ',(A),' field with bounds: (1:',I0,',1:',I0,'), internal region is (', " // &
            "I0,':',I0,',',I0,':',I0,')')") trim(adjustl(fld_type)), grid%nx, grid%ny, &
            self%internal%xstart, self%internal%xstop, self%internal%ystart, self%internal%ystop
    end if

    ! every field has the extents of the grid, whatever its point type (field_mod.f90:327-351)
    allocate(self%data(1:grid%nx, 1:grid%ny), stat=ierr)
    if (ierr /= 0) call gocean_stop('r2d_field_constructor: ERROR: failed to allocate field')
    self%data(:,:) = 0.0_go_wp

    if (present(init_global_data)) then
       dx = grid%subdomain%global%xstart - self%internal%xstart
       dy = grid%subdomain%global%ystart - self%internal%ystart
       do jj = grid%subdomain%internal%ystart, grid%subdomain%internal%ystop
          do ji = grid%subdomain%internal%xstart, grid%subdomain%internal%xstop
             self%data(ji, jj) = init_global_data(ji + dx, jj + dy)
          end do
       end do
    end if
  end function r2d_field_constructor

  !> Release the host array and, when it is one of ours, the device copy.
  subroutine r2d_free_field(fld)
    type(r2d_field), intent(inout) :: fld
    integer(c_int) :: rc
    if (allocated(fld%data)) deallocate(fld%data)
    if (c_associated(fld%device_ptr) .and. field_on_dlesm_device(fld)) then
       rc = dlesm_field_destroy(fld%device_ptr)
       fld%device_ptr = c_null_ptr
       fld%data_on_device = .false.
    end if
  end subroutine r2d_free_field

  ! ---------------------------------------------------------------------------
  ! device residency

  !> True when the field's device copy is managed by this library (as opposed to a foreign
  !! device model that registered its own callbacks).
  logical function field_on_dlesm_device(fld)
    class(r2d_field), intent(in) :: fld
    field_on_dlesm_device = associated(fld%read_from_device_c, dlesm_read_cb)
  end function field_on_dlesm_device

  !> The library's transfer callbacks in the exact (non-BIND(C)) shape of the reference's
  !! read_from_device_c_interface / write_to_device_c_interface; they forward to the C ABI.
  subroutine dlesm_read_cb(from, to, startx, starty, nx, ny, blocking)
    type(c_ptr), intent(in), value :: from
    type(c_ptr), intent(in), value :: to
    integer(c_int), intent(in), value :: startx, starty, nx, ny
    logical(c_bool), intent(in), value :: blocking
    call dlesm_read_from_device(from, to, startx, starty, nx, ny, blocking)
  end subroutine dlesm_read_cb

  subroutine dlesm_write_cb(from, to, startx, starty, nx, ny, blocking)
    type(c_ptr), intent(in), value :: from
    type(c_ptr), intent(in), value :: to
    integer(c_int), intent(in), value :: startx, starty, nx, ny
    logical(c_bool), intent(in), value :: blocking
    call dlesm_write_to_device(from, to, startx, starty, nx, ny, blocking)
  end subroutine dlesm_write_cb

  subroutine ensure_descriptor(fld)
    class(r2d_field), intent(inout) :: fld
    integer(c_int) :: rc
    if (c_associated(fld%device_ptr)) return
    rc = dlesm_field_create(int(size(fld%data, 1), c_int), int(size(fld%data, 2), c_int), fld%device_ptr)
    if (rc /= 0) call gocean_stop('field device buffer: ' // dlesm_error_text())
  end subroutine ensure_descriptor

  !> Move a field to the GPU: allocate its HBM copy, upload the host data, register the
  !! library's transfer callbacks and mark the device copy as the valid one.  This is what a
  !! PSy layer does before its first kernel launch on the field.
  subroutine field_to_device(fld)
    type(r2d_field), intent(inout), target :: fld
    if (fld%data_on_device) return
    call ensure_descriptor(fld)
    fld%read_from_device_c => dlesm_read_cb
    fld%write_to_device_c => dlesm_write_cb
    fld%data_on_device = .true.
    call fld%write_to_device()
  end subroutine field_to_device

  !> Raw device address of the field's data, for kernel launches through the C ABI.
  function field_device_data(fld) result(p)
    type(r2d_field), intent(in) :: fld
    type(c_ptr) :: p
    p = c_null_ptr
    if (c_associated(fld%device_ptr)) p = dlesm_field_data(fld%device_ptr)
  end function field_device_data

  subroutine patch_args(self, startx, starty, nx, ny, blocking, lx, ly, lnx, lny, lblock)
    class(r2d_field), intent(in) :: self
    integer, optional, intent(in) :: startx, starty, nx, ny
    logical, optional, intent(in) :: blocking
    integer, intent(out) :: lx, ly, lnx, lny
    logical, intent(out) :: lblock
    lx = 1;  ly = 1
    lnx = size(self%data, 1);  lny = size(self%data, 2)
    lblock = .true.
    if (present(startx)) lx = startx
    if (present(starty)) ly = starty
    if (present(nx)) lnx = nx
    if (present(ny)) lny = ny
    if (present(blocking)) lblock = blocking
  end subroutine patch_args

  !> Bring (a patch of) the device copy to the host; no-op for host-resident fields.
  subroutine read_from_device(self, startx, starty, nx, ny, blocking)
    class(r2d_field), target :: self
    integer, optional, intent(in) :: startx, starty, nx, ny
    logical, optional, intent(in) :: blocking
    integer :: lx, ly, lnx, lny
    logical :: lblock
    if (.not. self%data_on_device) return
    call patch_args(self, startx, starty, nx, ny, blocking, lx, ly, lnx, lny, lblock)
    if (associated(self%read_from_device_c)) then
       call self%read_from_device_c(self%device_ptr, c_loc(self%data), int(lx, c_int), int(ly, c_int), &
                                    int(lnx, c_int), int(lny, c_int), logical(lblock, c_bool))
    else if (associated(self%read_from_device_f)) then
       call self%read_from_device_f(self%device_ptr, self%data, lx, ly, lnx, lny, lblock)
    else
       call gocean_stop('ERROR: Data is on a device but no instructions about how to retrieve ' // &
                        'the data have been provided.')
    end if
  end subroutine read_from_device

  !> Push (a patch of) the host data to the device copy; no-op for host-resident fields.
  subroutine write_to_device(self, startx, starty, nx, ny, blocking)
    class(r2d_field), target :: self
    integer, optional, intent(in) :: startx, starty, nx, ny
    logical, optional, intent(in) :: blocking
    integer :: lx, ly, lnx, lny
    logical :: lblock
    if (.not. self%data_on_device) return
    call patch_args(self, startx, starty, nx, ny, blocking, lx, ly, lnx, lny, lblock)
    if (associated(self%write_to_device_c)) then
       call self%write_to_device_c(c_loc(self%data), self%device_ptr, int(lx, c_int), int(ly, c_int), &
                                   int(lnx, c_int), int(lny, c_int), logical(lblock, c_bool))
    else if (associated(self%write_to_device_f)) then
       call self%write_to_device_f(self%data, self%device_ptr, lx, ly, lnx, lny, lblock)
    else
       call gocean_stop('ERROR: Data is on a device but no instructions about how to write new ' // &
                        'data have been provided.')
    end if
  end subroutine write_to_device

  function get_data(self) result(dptr)
    class(r2d_field), target :: self
    real(go_wp), dimension(:,:), pointer :: dptr
    call self%read_from_device()
    dptr => self%data
  end function get_data

  function set_data(self, array) result(flag)
    class(r2d_field) :: self
    integer :: flag
    real(go_wp), dimension(:,:) :: array
    self%data = array
    call self%write_to_device()
    flag = 0
  end function set_data

  ! ---------------------------------------------------------------------------
  ! iteration bounds

  subroutine set_field_bounds(fld, fld_type, grid_points)
    class(field_type), intent(inout) :: fld
    integer, intent(in) :: grid_points
    character(len=8), intent(out) :: fld_type
    character(len=5), parameter :: names(0:4) = (/ 'C-U  ', 'C-V  ', 'C-T  ', 'C-F  ', 'C-All' /)
    type(c_region) :: sub, cin, cwh
    integer(c_int) :: rc

    if (grid_points < GO_U_POINTS .or. grid_points > GO_ALL_POINTS) then
       call gocean_stop('r2d_field_constructor: ERROR: invalid specifier for type of mesh points')
    end if
    fld_type = names(grid_points)
    fld%defined_on = grid_points
    associate (s => fld%grid%subdomain%internal)
      sub = c_region(s%nx, s%ny, s%xstart, s%xstop, s%ystart, s%ystop)
    end associate
    rc = dlesm_field_bounds(int(grid_points, c_int), int(fld%grid%offset, c_int), &
                            int(fld%grid%boundary_conditions(1), c_int), &
                            int(fld%grid%boundary_conditions(2), c_int), sub, &
                            int(fld%grid%nx, c_int), int(fld%grid%ny, c_int), cin, cwh)
    if (rc /= 0) call gocean_stop(dlesm_error_text())   ! the combinations the reference stops on
    fld%internal = from_c(cin)
    fld%whole = from_c(cwh)

    ! periodic boundaries of SW-offset grids are applied by patch copies between these
    ! source/destination regions (field_mod.f90:1394-1464)
    fld%num_halos = 0
    if (grid_points /= GO_ALL_POINTS .and. fld%grid%offset == GO_OFFSET_SW) call init_periodic_bc_halos(fld)
  end subroutine set_field_bounds

  function from_c(c) result(r)
    type(c_region), intent(in) :: c
    type(region_type) :: r
    r%nx = c%nx;  r%ny = c%ny
    r%xstart = c%xstart;  r%xstop = c%xstop
    r%ystart = c%ystart;  r%ystop = c%ystop
  end function from_c

  subroutine init_periodic_bc_halos(fld)
    class(field_type), intent(inout) :: fld
    integer :: ihalo
    logical :: px, py
    px = fld%grid%boundary_conditions(1) == GO_BC_PERIODIC
    py = fld%grid%boundary_conditions(2) == GO_BC_PERIODIC
    fld%num_halos = merge(2, 0, px) + merge(2, 0, py)
    if (allocated(fld%halo)) deallocate(fld%halo)
    allocate(fld%halo(fld%num_halos))
    ihalo = 0
    associate (it => fld%internal)
      if (px) then
         ! east halo column <- west-most internal column, then the reverse
         call set_halo(fld%halo(ihalo + 1), it%xstart, it%xstart, it%ystart, it%ystop, &
                       it%xstop + 1, it%xstop + 1, it%ystart, it%ystop)
         call set_halo(fld%halo(ihalo + 2), it%xstop, it%xstop, it%ystart, it%ystop, &
                       it%xstart - 1, it%xstart - 1, it%ystart, it%ystop)
         ihalo = ihalo + 2
      end if
      if (py) then
         ! north halo row <- south-most internal row (halo columns included), then the reverse
         call set_halo(fld%halo(ihalo + 1), it%xstart - 1, it%xstop + 1, it%ystart, it%ystart, &
                       it%xstart - 1, it%xstop + 1, it%ystop + 1, it%ystop + 1)
         call set_halo(fld%halo(ihalo + 2), it%xstart - 1, it%xstop + 1, it%ystop, it%ystop, &
                       it%xstart - 1, it%xstop + 1, it%ystart - 1, it%ystart - 1)
      end if
    end associate
  contains
    subroutine set_halo(h, sx0, sx1, sy0, sy1, dx0, dx1, dy0, dy1)
      type(halo_type), intent(inout) :: h
      integer, intent(in) :: sx0, sx1, sy0, sy1, dx0, dx1, dy0, dy1
      h%needs_update = 0
      h%source%xstart = sx0;  h%source%xstop = sx1
      h%source%ystart = sy0;  h%source%ystop = sy1
      h%source%nx = sx1 - sx0 + 1;  h%source%ny = sy1 - sy0 + 1
      h%dest%xstart = dx0;  h%dest%xstop = dx1
      h%dest%ystart = dy0;  h%dest%ystop = dy1
      h%dest%nx = dx1 - dx0 + 1;  h%dest%ny = dy1 - dy0 + 1
    end subroutine set_halo
  end subroutine init_periodic_bc_halos

  ! ---------------------------------------------------------------------------
  ! copies, fills, checksum

  subroutine copy_2dfield_array(field_in, field_out)
    real(go_wp), intent(in), dimension(:,:) :: field_in
    real(go_wp), intent(out), dimension(:,:) :: field_out
    field_out(:,:) = field_in(:,:)
  end subroutine copy_2dfield_array

  subroutine copy_2dfield_array_patch(field, src, dest)
    real(go_wp), intent(inout), dimension(:,:) :: field
    type(region_type), intent(in) :: src, dest
    field(dest%xstart:dest%xstop, dest%ystart:dest%ystop) = &
         field(src%xstart:src%xstop, src%ystart:src%ystop)
  end subroutine copy_2dfield_array_patch

  !> Whole-field copy; device to device when both fields live there.
  subroutine copy_2dfield(field_in, field_out)
    type(r2d_field), intent(in) :: field_in
    type(r2d_field), intent(inout) :: field_out
    integer(c_int) :: rc
    if (field_in%data_on_device .and. field_out%data_on_device .and. &
        field_on_dlesm_device(field_in) .and. field_on_dlesm_device(field_out)) then
       rc = dlesm_copy_patch_f64(field_device_data(field_in), field_device_data(field_out), &
                                 int(size(field_in%data, 1), c_int), int(size(field_in%data, 2), c_int), &
                                 1_c_int, 1_c_int, 1_c_int, 1_c_int, &
                                 int(size(field_in%data, 1), c_int), int(size(field_in%data, 2), c_int), c_null_ptr)
       if (rc /= 0) call gocean_stop('copy_field: ' // dlesm_error_text())
    else
       field_out%data(:,:) = field_in%data(:,:)
    end if
  end subroutine copy_2dfield

  !> Patch copy inside one field (how periodic boundaries are applied).
  subroutine copy_2dfield_patch(field, src, dest)
    type(r2d_field), intent(inout) :: field
    type(region_type), intent(in) :: src, dest
    integer(c_int) :: rc
    if (field%data_on_device .and. field_on_dlesm_device(field)) then
       rc = dlesm_copy_patch_f64(field_device_data(field), field_device_data(field), &
                                 int(size(field%data, 1), c_int), int(size(field%data, 2), c_int), &
                                 int(src%xstart, c_int), int(src%ystart, c_int), &
                                 int(dest%xstart, c_int), int(dest%ystart, c_int), &
                                 int(src%xstop - src%xstart + 1, c_int), &
                                 int(src%ystop - src%ystart + 1, c_int), c_null_ptr)
       if (rc /= 0) call gocean_stop('copy_field: ' // dlesm_error_text())
    else
       field%data(dest%xstart:dest%xstop, dest%ystart:dest%ystop) = &
            field%data(src%xstart:src%xstop, src%ystart:src%ystop)
    end if
  end subroutine copy_2dfield_patch

  subroutine set_field(fld, val)
    class(field_type), intent(inout) :: fld
    real(go_wp), intent(in) :: val
    integer(c_int) :: rc
    select type (fld)
    type is (r2d_field)
       fld%data = val
       if (fld%data_on_device .and. field_on_dlesm_device(fld)) then
          rc = dlesm_fill_f64(field_device_data(fld), int(size(fld%data, 1), c_int), &
                              int(size(fld%data, 2), c_int), 1_c_int, int(size(fld%data, 1), c_int), &
                              1_c_int, int(size(fld%data, 2), c_int), real(val, c_double), c_null_ptr)
          if (rc /= 0) call gocean_stop('set_field: ' // dlesm_error_text())
       else
          call fld%write_to_device()
       end if
    class default
    end select
  end subroutine set_field

  !> SUM(ABS()) of the internal region, summed over all ranks.
  function fld_checksum(field) result(val)
    use parallel_comms_mod, only: global_sum
    type(r2d_field), intent(in), target :: field
    real(go_wp) :: val
    real(c_double) :: dval
    integer(c_int) :: rc
    if (field%data_on_device .and. field_on_dlesm_device(field)) then
       rc = dlesm_checksum_f64(field_device_data(field), int(size(field%data, 1), c_int), &
                               int(size(field%data, 2), c_int), int(field%internal%xstart, c_int), &
                               int(field%internal%xstop, c_int), int(field%internal%ystart, c_int), &
                               int(field%internal%ystop, c_int), dval, c_null_ptr)
       if (rc /= 0) call gocean_stop('field_checksum: ' // dlesm_error_text())
       val = dval
       call global_sum(val)
    else
       call sync_host(field)
       val = array_checksum(field%data, field%internal%xstart, field%internal%xstop, &
                            field%internal%ystart, field%internal%ystop)
    end if
  contains
    subroutine sync_host(f)      ! get_data() on an intent(in) object
      type(r2d_field), intent(in), target :: f
      type(r2d_field), pointer :: p
      real(go_wp), pointer :: d(:,:)
      p => f
      d => p%get_data()
    end subroutine sync_host
  end function fld_checksum

  function array_checksum(field, xstart, xstop, ystart, ystop) result(val)
    use parallel_comms_mod, only: global_sum
    real(go_wp), dimension(:,:), intent(in) :: field
    integer, optional, intent(in) :: xstart, xstop, ystart, ystop
    real(go_wp) :: val
    if (present(xstart)) then
       val = sum(abs(field(xstart:xstop, ystart:ystop)))
    else
       val = sum(abs(field(:,:)))
    end if
    call global_sum(val)
  end function array_checksum

  ! ---------------------------------------------------------------------------
  ! halo exchange

  !> Depth-1 halo swap with the neighbouring subdomains. `depth` is ignored as in the
  !! reference (field_mod.f90:1226-1228).
  subroutine halo_exchange(self, depth)
    use parallel_comms_mod, only: Iplus, Iminus, Jplus, Jminus, exchange_generic, exchange_device, &
                                  nsend, nrecv, isrcsend, jsrcsend, nxsend, nysend, idesrecv, &
                                  jdesrecv, nxrecv, nyrecv
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    class(r2d_field), target, intent(inout) :: self
    integer, intent(in) :: depth
    integer :: exch, k

    if (.not. DIST_MEM_ENABLED) return
    if (self%data_on_device .and. field_on_dlesm_device(self)) then
       ! strips go GPU to GPU over RCCL; the host copy is not touched
       call exchange_device(field_device_data(self), size(self%data, 1), size(self%data, 2), &
                            Jplus, Jminus, Iplus, Iminus)
    else
       ! a foreign device model: fetch what every message sends, exchange on the host side,
       ! push back what every message received (all nsend/nrecv messages, including the
       ! diagonal ones the reference's four fixed calls miss, SURVEY.md section 8a H3)
       if (self%data_on_device) then
          do k = 1, nsend
             call self%read_from_device(isrcsend(k), jsrcsend(k), nxsend(k), nysend(k), k == nsend)
          end do
       end if
       call exchange_generic(b2=self%data, handle=exch, comm1=Jplus, comm2=Jminus, comm3=Iplus, comm4=Iminus)
       if (self%data_on_device) then
          do k = 1, nrecv
             call self%write_to_device(idesrecv(k), jdesrecv(k), nxrecv(k), nyrecv(k), k == nrecv)
          end do
       end if
    end if
  end subroutine halo_exchange

  subroutine read_halo_from_device(self, comm, blocking)
    use parallel_comms_mod, only: isrcsend, jsrcsend, nxsend, nysend
    class(r2d_field), target, intent(inout) :: self
    integer, intent(in) :: comm
    logical, intent(in) :: blocking
    if (isrcsend(comm) > 0) then
       call self%read_from_device(isrcsend(comm), jsrcsend(comm), nxsend(comm), nysend(comm), blocking)
    end if
  end subroutine read_halo_from_device

  subroutine write_halo_to_device(self, comm, blocking)
    use parallel_comms_mod, only: idesrecv, jdesrecv, nxrecv, nyrecv
    class(r2d_field), target, intent(inout) :: self
    integer, intent(in) :: comm
    logical, intent(in) :: blocking
    if (idesrecv(comm) > 0) then
       call self%write_to_device(idesrecv(comm), jdesrecv(comm), nxrecv(comm), nyrecv(comm), blocking)
    end if
  end subroutine write_halo_to_device

  !> Collect the internal regions of all ranks into one global array on the master rank.
  subroutine gather_inner_data(self, global_data)
    use parallel_utils_mod, only: get_num_ranks, gather
    use parallel_mod, only: on_master
    class(r2d_field), intent(in), target :: self
    real(go_wp), dimension(:,:), allocatable, intent(out) :: global_data
    real(go_wp), dimension(:), allocatable, target :: send_buffer, recv_buffer
    type(r2d_field), pointer :: me
    real(go_wp), pointer :: d(:,:)
    integer :: n, r, w, h, ierr, halo_x, halo_y

    allocate(global_data(self%grid%global_nx, self%grid%global_ny), stat=ierr)
    if (ierr /= 0) call gocean_stop('gather_inner_data failed to allocate global result array')
    me => self
    if (self%data_on_device .and. field_on_dlesm_device(me)) then
       ! device-resident field: pack, gather (RCCL) and unpack on the device, one copy of the
       ! assembled global array to the host -- no full-field download, no host loops
       call gather_on_device(me, global_data)
       return
    end if
    d => me%get_data()          ! host copy up to date
    associate (it => self%internal)
      if (get_num_ranks() == 1) then
         global_data(1:it%nx, 1:it%ny) = d(it%xstart:it%xstop, it%ystart:it%ystop)
         return
      end if
      ! fixed-size slots: the largest tile without its halos (field_mod.f90:1348-1351)
      halo_x = it%xstart - 1;  halo_y = it%ystart - 1
      n = (self%grid%decomp%max_width - 2 * halo_x) * (self%grid%decomp%max_height - 2 * halo_y)
      allocate(send_buffer(n), recv_buffer(n * get_num_ranks()), stat=ierr)
      if (ierr /= 0) call gocean_stop('gather_inner_data failed to allocate buffers')
      send_buffer = 0.0_go_wp
      send_buffer(1:it%nx * it%ny) = reshape(d(it%xstart:it%xstop, it%ystart:it%ystop), (/ it%nx * it%ny /))
    end associate
    call gather(send_buffer, recv_buffer)
    if (on_master()) then
       do r = 1, get_num_ranks()
          associate (g => self%grid%decomp%subdomains(r)%global)
            w = g%xstop - g%xstart + 1;  h = g%ystop - g%ystart + 1
            global_data(g%xstart:g%xstop, g%ystart:g%ystop) = &
                 reshape(recv_buffer((r - 1) * n + 1:(r - 1) * n + w * h), (/ w, h /))
          end associate
       end do
    end if
  end subroutine gather_inner_data

  subroutine gather_on_device(fld, global_data)
    use parallel_utils_mod, only: get_num_ranks
    use parallel_comms_mod, only: to_c_decomp
    type(r2d_field), intent(in), target :: fld
    real(go_wp), dimension(:,:), intent(inout), target, contiguous :: global_data
    type(c_decomp) :: cd
    type(c_subdomain), allocatable :: csubs(:)
    type(c_region) :: cint
    integer(c_int) :: rc
    call to_c_decomp(fld%grid%decomp, cd, csubs)
    associate (it => fld%internal)
      cint = c_region(it%nx, it%ny, it%xstart, it%xstop, it%ystart, it%ystop)
    end associate
    rc = dlesm_gather_inner_f64(field_device_data(fld), int(size(fld%data, 1), c_int), &
                                int(size(fld%data, 2), c_int), cint, cd, csubs, &
                                int(get_num_ranks(), c_int), c_loc(global_data))
    if (rc /= 0) call gocean_stop('gather_inner_data: ' // dlesm_error_text())
  end subroutine gather_on_device

  !> GOCEAN_OMP_GRID="NxM": dimensions of the OpenMP tiling grid (field_mod.f90:1473-1503)
  function get_grid_dims(nx, ny) result(success)
    integer, intent(inout) :: nx, ny
    logical :: success
    character(len=20) :: lstr
    integer :: idx, ierr
    success = .false.
    call get_environment_variable(name='GOCEAN_OMP_GRID', value=lstr, status=ierr)
    if (ierr /= 0) return
    idx = index(lstr, 'x')
    if (idx == 0) then
       write (*, "(/'get_grid_dims: failed to parse GOCEAN_OMP_GRID string: ',(A))") trim(lstr)
       return
    end if
    read(lstr(1:idx-1), *, iostat=ierr) nx
    if (ierr /= 0) return
    read(lstr(idx+1:), *, iostat=ierr) ny
    success = ierr == 0
  end function get_grid_dims

end module field_mod
