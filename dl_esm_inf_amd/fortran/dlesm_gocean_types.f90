!> Small definitional modules of the dl_esm_inf API, kept in one file:
!!   kind_params_mod, region_mod, tile_mod, halo_mod, decomposition_mod,
!!   global_parameters_mod, argument_mod, kernel_mod
!! Module, type, component and constant names (and the constants' values) are
!! those of the reference so that GOcean / PSyclone-generated code compiles
!! unchanged against this library; reference locations are cited per module
!! (paths relative to finite_difference/src/).

!> Working precision. Reference: kind_params_mod.f90:9,12 (fp64 everywhere).
module kind_params_mod
  use iso_c_binding
  implicit none
  public
  integer, parameter :: go_wp = selected_real_kind(12, 307)
  integer, parameter :: go_dp = c_double
end module kind_params_mod

!> A rectangular index region. Reference: region_mod.f90:7-32.
module region_mod
  implicit none
  type :: region_type
     integer :: nx, ny
     integer :: xstart, xstop
     integer :: ystart, ystop
  end type region_type
  interface region_type
     module procedure new_region
  end interface region_type
contains
  function new_region() result(r)
    type(region_type) :: r
    r%nx = 0;  r%ny = 0
    r%xstart = 0;  r%xstop = 0
    r%ystart = 0;  r%ystop = 0
  end function new_region
end module region_mod

!> Internal/whole regions of one OpenMP tile. Reference: tile_mod.f90:36-42.
module tile_mod
  use region_mod
  implicit none
  type :: tile_type
     type(region_type) :: internal
     type(region_type) :: whole
  end type tile_type
end module tile_mod

!> Source/destination patch of a (periodic-boundary) halo. Reference: halo_mod.f90:9-25.
module halo_mod
  use region_mod
  implicit none
  private
  type, public :: halo_type
     integer :: needs_update
     type(region_type) :: source
     type(region_type) :: dest
  end type halo_type
end module halo_mod

!> Regular 2-D block decomposition. Reference: decomposition_mod.f90:44-68.
!! subdomain%global is the position of the internal part in the global domain
!! while global%nx/ny hold the WHOLE local extent (internal + halos).
module decomposition_mod
  use region_mod, only: region_type
  implicit none
  type :: subdomain_type
     type(region_type) :: global
     type(region_type) :: internal
  end type subdomain_type
  type :: decomposition_type
     integer :: global_nx, global_ny
     integer :: nx, ny
     integer :: ndomains
     integer :: max_width, max_height
     type(subdomain_type), allocatable :: subdomains(:)
     integer, allocatable :: proc_subdomains(:,:)
  end type decomposition_type
end module decomposition_mod

!> Reference: global_parameters_mod.f90:9-23.
module global_parameters_mod
  use iso_c_binding
  implicit none
  private
  integer, parameter, public :: NAME_LEN = 1024
  public :: GO_CELLS, GO_EDGES, GO_VERTICES, GO_FE
  enum, bind(c)
     enumerator :: GO_CELLS = 2, GO_EDGES = 1, GO_VERTICES = 0
  end enum
  enum, bind(c)
     enumerator :: GO_FE
  end enum
end module global_parameters_mod

!> Kernel-argument metadata vocabulary read by PSyclone at code-generation time.
!! Reference: argument_mod.f90:39-112. No run-time behaviour.
module argument_mod
  use iso_c_binding
  use global_parameters_mod
  implicit none
  private

  enum, bind(c)
     enumerator :: GO_READ
     enumerator :: GO_WRITE, GO_READWRITE, GO_INC
     enumerator :: GO_MIN, GO_MAX, GO_SUM
  end enum
  public :: GO_READ, GO_WRITE, GO_READWRITE, GO_INC, GO_MIN, GO_MAX, GO_SUM

  !> 3x3 stencil encoded as three integers, one per row
  type, public :: go_stencil
     integer :: first_row
     integer :: second_row
     integer :: third_row
  end type go_stencil

  type, public :: go_arg
     integer :: arg_intent
     integer :: element
     type(go_stencil) :: stencil_type = go_stencil(0, 0, 0)
  end type go_arg

  integer, public, parameter :: GO_R_SCALAR = 0, GO_I_SCALAR = 1
  integer, public, parameter :: GO_EVERY = 1
  integer, public, parameter :: GO_CU = 1, GO_CV = 2, GO_CT = 3, GO_CF = 4
  ! grid properties a kernel may request
  integer, public, parameter :: GO_TIME_STEP = 1
  integer, public, parameter :: GO_GRID_AREA_T = 2, GO_GRID_AREA_U = 3, GO_GRID_AREA_V = 4
  integer, public, parameter :: GO_GRID_MASK_T = 5
  integer, public, parameter :: GO_GRID_DX_T = 6, GO_GRID_DX_U = 7, GO_GRID_DX_V = 8
  integer, public, parameter :: GO_GRID_DY_T = 9, GO_GRID_DY_U = 10, GO_GRID_DY_V = 11
  integer, public, parameter :: GO_GRID_LAT_U = 12, GO_GRID_LAT_V = 13
  integer, public, parameter :: GO_GRID_DX_CONST = 14, GO_GRID_DY_CONST = 15
  integer, public, parameter :: GO_GRID_X_MIN_INDEX = 16, GO_GRID_X_MAX_INDEX = 17
  integer, public, parameter :: GO_GRID_Y_MIN_INDEX = 18, GO_GRID_Y_MAX_INDEX = 19
end module argument_mod

!> Base type of kernel metadata. Reference: kernel_mod.f90:28-50.
module kernel_mod
  use argument_mod
  use global_parameters_mod
  implicit none
  private
  public :: GO_CELLS, GO_EDGES, GO_VERTICES, GO_FE, GO_ARG
  public :: GO_READ, GO_WRITE, GO_READWRITE, GO_INC
  public :: GO_SUM, GO_MIN, GO_MAX
  integer, public, parameter :: GO_DOFS = 5
  type(go_stencil), public, parameter :: GO_POINTWISE = go_stencil(000, 010, 000)
  integer, public, parameter :: GO_INTERNAL_PTS = 0, GO_EXTERNAL_PTS = 1, GO_ALL_PTS = 2
  integer, public, parameter :: GO_ORTHOGONAL_REGULAR = 7, GO_ORTHOGONAL_CURVILINEAR = 8
  type, public :: kernel_type
     private
     logical :: no_op
  end type kernel_type
end module kernel_mod
