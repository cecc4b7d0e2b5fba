!> The finite-difference grid object of the dl_esm_inf API.
!! Type, component, constant and procedure names follow the reference
!! (finite_difference/src/grid_mod.f90:45-163) because generated PSy layers touch
!! the components directly (grid%nx, grid%subdomain%internal%xstart, grid%dx_t ...).
!! Extents and message tables come from the C-ABI library; the metric arrays are the
!! reference's constant fills for a regular orthogonal mesh.
module grid_mod
  use iso_c_binding, only: c_ptr, c_null_ptr, c_int
  use kind_params_mod
  use region_mod
  use gocean_mod
  use decomposition_mod, only: subdomain_type, decomposition_type
  implicit none
  private

  integer, public, parameter :: GO_ARAKAWA_C = 0, GO_ARAKAWA_B = 1
  ! how U/V/F indices are offset from the T point (grid_mod.f90:48-60)
  integer, public, parameter :: GO_OFFSET_SW = 0, GO_OFFSET_SE = 1, GO_OFFSET_NW = 2, &
                                GO_OFFSET_NE = 3, GO_OFFSET_ANY = 4
  integer, public, parameter :: GO_BC_PERIODIC = 0, GO_BC_EXTERNAL = 1, GO_BC_NONE = 2
  integer, parameter :: HALO_WIDTH_X = 1, HALO_WIDTH_Y = 1

  type, public :: grid_type
     integer :: name
     integer :: offset
     integer :: global_nx, global_ny
     !> allocated extents of every field on this grid (nx is padded to DL_ESM_ALIGNMENT)
     integer :: nx
     integer :: ny
     real(go_wp) :: dx
     real(go_wp) :: dy
     !> 1 wet inside the simulated region, 0 land, -1 wet outside it
     integer, allocatable :: tmask(:,:)
     type(c_ptr) :: tmask_device
     integer, dimension(3) :: boundary_conditions
     type(subdomain_type) :: subdomain
     type(decomposition_type) :: decomp
     real(go_wp), allocatable :: dx_t(:,:), dy_t(:,:)
     type(c_ptr) :: dx_t_device, dy_t_device
     real(go_wp), allocatable :: dx_u(:,:), dy_u(:,:)
     type(c_ptr) :: dx_u_device, dy_u_device
     real(go_wp), allocatable :: dx_v(:,:), dy_v(:,:)
     type(c_ptr) :: dx_v_device, dy_v_device
     real(go_wp), allocatable :: dx_f(:,:), dy_f(:,:)
     type(c_ptr) :: dx_f_device, dy_f_device
     real(go_wp), allocatable :: area_t(:,:), area_u(:,:), area_v(:,:)
     type(c_ptr) :: area_t_device, area_u_device, area_v_device
     real(go_wp), allocatable :: gphiu(:,:)
     real(go_wp), allocatable :: gphiv(:,:)
     real(go_wp), allocatable :: gphif(:,:)
     type(c_ptr) :: gphiu_device, gphiv_device, gphif_device
     real(go_wp), allocatable :: xt(:,:), yt(:,:)
     type(c_ptr) :: xt_device, yt_device
   contains
     procedure :: get_tmask
     procedure :: decompose
  end type grid_type

  interface grid_type
     module procedure grid_constructor
  end interface grid_type

  public grid_init, HALO_WIDTH_X, HALO_WIDTH_Y

contains

  function get_tmask(self) result(tmask)
    class(grid_type), target, intent(in) :: self
    integer, pointer :: tmask(:,:)
    tmask => self%tmask
  end function get_tmask

  !> Decompose a domainx x domainy domain over the ranks and keep this rank's tile.
  subroutine decompose(self, domainx, domainy, ndomains, ndomainx, ndomainy, halo_width)
    use parallel_mod, only: get_rank, go_decompose
    class(grid_type), target, intent(inout) :: self
    integer, intent(in) :: domainx, domainy
    integer, intent(in), optional :: ndomains, ndomainx, ndomainy, halo_width
    self%decomp = go_decompose(domainx, domainy, ndomains=ndomains, ndomainx=ndomainx, &
                               ndomainy=ndomainy, halo_width=halo_width)
    self%subdomain = self%decomp%subdomains(get_rank())
    self%global_nx = self%decomp%global_nx
    self%global_ny = self%decomp%global_ny
  end subroutine decompose

  function grid_constructor(grid_name, boundary_conditions, grid_offsets) result(self)
    integer, intent(in) :: grid_name
    integer, dimension(3), intent(in) :: boundary_conditions
    integer, optional, intent(in) :: grid_offsets
    type(grid_type), target :: self

    if (.not. present(grid_offsets)) then
       call gocean_stop('ERROR: grid offset not specified in call to grid_constructor.')
    end if
    if (grid_name /= GO_ARAKAWA_C .and. grid_name /= GO_ARAKAWA_B) then
       write(*, *) 'grid_constructor: ERROR: unsupported grid type: ', grid_name
       call gocean_stop('')
    end if
    if (grid_offsets < GO_OFFSET_SW .or. grid_offsets > GO_OFFSET_NE) then
       write(*, *) 'grid_constructor: ERROR: unsupported relative offsets of grid types: ', grid_offsets
       call gocean_stop('')
    end if
    self%name = grid_name
    self%offset = grid_offsets
    self%boundary_conditions(1:3) = boundary_conditions(1:3)
    self%nx = 0;  self%ny = 0
    ! no device mirrors yet
    self%tmask_device = c_null_ptr
    self%dx_t_device = c_null_ptr;  self%dy_t_device = c_null_ptr
    self%dx_u_device = c_null_ptr;  self%dy_u_device = c_null_ptr
    self%dx_v_device = c_null_ptr;  self%dy_v_device = c_null_ptr
    self%dx_f_device = c_null_ptr;  self%dy_f_device = c_null_ptr
    self%area_t_device = c_null_ptr;  self%area_u_device = c_null_ptr
    self%area_v_device = c_null_ptr
    self%gphiu_device = c_null_ptr;  self%gphiv_device = c_null_ptr
    self%gphif_device = c_null_ptr
    self%xt_device = c_null_ptr;  self%yt_device = c_null_ptr
  end function grid_constructor

  !> Complete the grid once it has been decomposed: padded extents (DL_ESM_ALIGNMENT),
  !! T mask, metric arrays, and the halo-exchange message tables.
  subroutine grid_init(grid, dxarg, dyarg, tmask)
    use parallel_mod, only: map_comms, get_rank, get_num_ranks, on_master
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    use dlesm_hip_mod
    type(grid_type), intent(inout) :: grid
    real(go_wp), intent(in) :: dxarg, dyarg
    integer, allocatable, dimension(:,:), intent(in), optional :: tmask
    integer(c_int) :: alignment, cnx, cny, rc
    integer :: ierr, ji, jj, xstart, xstop, ystart, ystop
    logical :: periodic

    rc = dlesm_alignment_from_env(alignment)
    if (rc /= 0) call gocean_stop(dlesm_error_text())
    rc = dlesm_grid_extents(int(grid%subdomain%global%nx, c_int), int(grid%subdomain%global%ny, c_int), &
                            alignment, cnx, cny)
    if (rc /= 0) call gocean_stop('grid_init: ' // dlesm_error_text())
    grid%nx = cnx
    grid%ny = cny
    if (alignment > 1 .and. (on_master() .or. get_rank() == grid%decomp%nx)) then
       write(*, "('Rank',I3,' contiguous dimension is',I6,' (it has',I4,' padding elements to " // &
            "satisfy ',I4,'-wide alignment)')") get_rank(), grid%nx, &
            grid%nx - grid%subdomain%global%nx - 1, alignment
    end if

    xstart = grid%subdomain%internal%xstart;  xstop = grid%subdomain%internal%xstop
    ystart = grid%subdomain%internal%ystart;  ystop = grid%subdomain%internal%ystop
    periodic = grid%boundary_conditions(1) == GO_BC_PERIODIC .or. &
               grid%boundary_conditions(2) == GO_BC_PERIODIC

    allocate(grid%tmask(grid%nx, grid%ny), stat=ierr)
    if (ierr /= 0) call gocean_stop('grid_init: failed to allocate array for T mask')
    if (present(tmask)) then
       ! the supplied mask covers the subdomain incl. its boundary ring; replicate its edge
       ! values into the padding rows and columns (grid_mod.f90:400-431)
       grid%tmask(xstart-1:xstop+1, ystart-1:ystop+1) = tmask(xstart-1:xstop+1, ystart-1:ystop+1)
       do jj = ystop + 2, grid%ny
          grid%tmask(:, jj) = grid%tmask(:, ystop + 1)
       end do
       do jj = 1, ystart - 2
          grid%tmask(:, jj) = grid%tmask(:, ystart - 1)
       end do
       do ji = 1, xstart - 2
          grid%tmask(ji, :) = grid%tmask(xstart - 1, :)
       end do
       do ji = xstop + 2, grid%nx
          grid%tmask(ji, :) = grid%tmask(xstop + 1, :)
       end do
    else
       if (get_num_ranks() > 1 .and. periodic) then
          call gocean_stop('grid_init: ERROR: Periodic boundary conditions are not yet supported.')
       end if
       grid%tmask(xstart-1:xstop+1, ystart-1:ystop+1) = 1   ! all wet
    end if

    grid%dx = dxarg
    grid%dy = dyarg
    allocate(grid%dx_t(grid%nx, grid%ny), grid%dy_t(grid%nx, grid%ny), &
             grid%dx_u(grid%nx, grid%ny), grid%dy_u(grid%nx, grid%ny), &
             grid%dx_f(grid%nx, grid%ny), grid%dy_f(grid%nx, grid%ny), &
             grid%dx_v(grid%nx, grid%ny), grid%dy_v(grid%nx, grid%ny), &
             grid%area_t(grid%nx, grid%ny), grid%area_u(grid%nx, grid%ny), &
             grid%area_v(grid%nx, grid%ny), grid%gphiu(grid%nx, grid%ny), &
             grid%gphiv(grid%nx, grid%ny), grid%gphif(grid%nx, grid%ny), &
             grid%xt(grid%nx, grid%ny), grid%yt(grid%nx, grid%ny), stat=ierr)
    if (ierr /= 0) call gocean_stop('grid_init: failed to allocate arrays')

    ! regular orthogonal mesh: constant spacing, f-plane at 50 degrees
    grid%dx_t = grid%dx;  grid%dy_t = grid%dy
    grid%dx_u = grid%dx;  grid%dy_u = grid%dy
    grid%dx_v = grid%dx;  grid%dy_v = grid%dy
    grid%dx_f = grid%dx;  grid%dy_f = grid%dy
    grid%area_t = grid%dx * grid%dy
    grid%area_u = grid%dx * grid%dy
    grid%area_v = grid%dx * grid%dy
    grid%gphiu = 50._go_wp
    grid%gphiv = 50._go_wp
    grid%gphif = 50._go_wp
    ! T-point coordinates: the first internal column/row carries its global index times the
    ! spacing, the rest follows by repeated addition in both directions (grid_mod.f90:536-556;
    ! repeated addition, not multiplication, to reproduce the reference's rounding)
    grid%xt(xstart, :) = grid%subdomain%global%xstart * grid%dx
    grid%yt(:, ystart) = grid%subdomain%global%ystart * grid%dy
    do ji = xstart + 1, grid%nx
       grid%xt(ji, :) = grid%xt(ji - 1, :) + grid%dx
    end do
    do jj = ystart + 1, grid%ny
       grid%yt(:, jj) = grid%yt(:, jj - 1) + grid%dy
    end do
    do ji = xstart - 1, 1, -1
       grid%xt(ji, :) = grid%xt(ji + 1, :) - grid%dx
    end do
    do jj = ystart - 1, 1, -1
       grid%yt(:, jj) = grid%yt(:, jj + 1) - grid%dy
    end do

    if (DIST_MEM_ENABLED .and. periodic) then
       call gocean_stop('map_comms call needs to be implemented for periodic boundary conditions.')
    end if
    ! the reference passes HALO_WIDTH_X/Y = 1 whatever the decomposition (grid_mod.f90:72-73);
    ! here a decomposition made with halo_width d > 1 gets depth-d tables (see map_comms)
    call map_comms(grid%decomp, grid%tmask, .false., &
                   (/grid%subdomain%internal%xstart - 1, grid%subdomain%internal%ystart - 1/), ierr)
    if (ierr /= 0) call gocean_stop('Set-up of communication tables (call to map_comms()) failed.')
  end subroutine grid_init

end module grid_mod
