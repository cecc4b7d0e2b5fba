"""Python mirror of grid_mod (reference: finite_difference/src/grid_mod.f90)."""
import ctypes as C

from . import _cabi, parallel_mod
from ._cabi import check

GO_ARAKAWA_C, GO_ARAKAWA_B = 0, 1
GO_OFFSET_SW, GO_OFFSET_SE, GO_OFFSET_NW, GO_OFFSET_NE, GO_OFFSET_ANY = 0, 1, 2, 3, 4
GO_BC_PERIODIC, GO_BC_EXTERNAL, GO_BC_NONE = 0, 1, 2


class grid_type:
    """grid_type (grid_mod.f90:75-157): only what the hot path touches -- extents,
    decomposition, spacing and the halo-exchange tables.  The 16 metric arrays of the
    reference are constant fills (grid_mod.f90:479-556) and are not materialised here."""

    def __init__(self, grid_name, boundary_conditions, grid_offsets=None):
        if grid_offsets is None:
            raise _cabi.GoceanStop(_cabi.EABORT, "ERROR: grid offset not specified in call to "
                                                 "grid_constructor.")
        if grid_name not in (GO_ARAKAWA_C, GO_ARAKAWA_B):
            raise _cabi.GoceanStop(_cabi.EABORT, f"grid_constructor: ERROR: unsupported grid type: {grid_name}")
        if grid_offsets not in (GO_OFFSET_SW, GO_OFFSET_SE, GO_OFFSET_NW, GO_OFFSET_NE):
            raise _cabi.GoceanStop(_cabi.EABORT, "grid_constructor: ERROR: unsupported relative offsets "
                                                 f"of grid types: {grid_offsets}")
        self.name = grid_name
        self.offset = grid_offsets
        self.boundary_conditions = list(boundary_conditions[:3])
        self.decomp = None
        self.subdomain = None
        self.nx = self.ny = 0
        self.global_nx = self.global_ny = 0
        self.dx = self.dy = 0.0
        self.comm_tables = None
        self._halo_plan = None
        self.halo_width = 1

    def decompose(self, domainx, domainy, ndomains=None, ndomainx=None, ndomainy=None, halo_width=1):
        """grid_mod.f90:183-211"""
        self.decomp = parallel_mod.go_decompose(domainx, domainy, ndomains, ndomainx, ndomainy,
                                                halo_width)
        self.subdomain = self.decomp.subdomains[parallel_mod.get_rank() - 1]
        self.global_nx, self.global_ny = self.decomp.global_nx, self.decomp.global_ny
        self.halo_width = halo_width


def grid_init(grid, dxarg, dyarg, tmask=None):
    """grid_init (grid_mod.f90:330-570): padded extents from DL_ESM_ALIGNMENT, then the
    halo-exchange tables."""
    L = _cabi.lib()
    a = C.c_int()
    check(L.dlesm_alignment_from_env(C.byref(a)))
    nx, ny = C.c_int(), C.c_int()
    check(L.dlesm_grid_extents(grid.subdomain.glob.nx, grid.subdomain.glob.ny, a.value,
                               C.byref(nx), C.byref(ny)))
    grid.nx, grid.ny = nx.value, ny.value
    periodic = GO_BC_PERIODIC in grid.boundary_conditions[:2]
    nranks = parallel_mod.get_num_ranks()
    if tmask is None and nranks > 1 and periodic:          # grid_mod.f90:437-442
        raise _cabi.GoceanStop(_cabi.EABORT, "grid_init: ERROR: Periodic boundary conditions are "
                                             "not yet supported.")
    grid.dx, grid.dy = float(dxarg), float(dyarg)
    if nranks > 1:
        if periodic:                                       # grid_mod.f90:559-564
            raise _cabi.GoceanStop(_cabi.EABORT, "map_comms call needs to be implemented for "
                                                 "periodic boundary conditions.")
        # halo_width 1: the reference's tables.  A decomposition made with halo_width d > 1 gets
        # the depth-d extension (what the fused multi-step kernels need); the reference would
        # call map_comms with HALO_WIDTH_X/Y = 1 here whatever the width (grid_mod.f90:72-73).
        hw = getattr(grid, "halo_width", 1)
        grid.comm_tables = parallel_mod.map_comms(grid.decomp, depth=None if hw == 1 else hw)
    else:
        grid.comm_tables = _cabi.CommTables()              # serial: no messages (pcomms:216)


def halo_plan(grid):
    """one device message plan per grid: every field shares the grid's extents"""
    if grid._halo_plan is None:
        p = C.c_void_p()
        check(_cabi.lib().dlesm_halo_plan_create(C.byref(grid.comm_tables), grid.nx, grid.ny,
                                                 C.byref(p)))
        grid._halo_plan = p
    return grid._halo_plan
