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
        self.tmask = None              # host copy, (ny, nx) int32 (grid_mod.f90:100)
        self._tmask_device = None      # its HBM mirror (grid_mod.f90:104-106), made on first use
        self._area_t_device = None

    @property
    def tmask_device(self):
        """device mirror of the T mask, a (ny, nx) int32 tensor"""
        if self._tmask_device is None:
            import torch
            if self.tmask is None:
                raise _cabi.GoceanStop(_cabi.EABORT, "grid%tmask requested before grid_init")
            self._tmask_device = torch.from_numpy(self.tmask).cuda()
            torch.cuda.current_stream().synchronize()      # in place before a kernel on another stream reads it
        return self._tmask_device

    @property
    def tmask_device_ptr(self):
        return C.c_void_p(self.tmask_device.data_ptr())

    @property
    def area_t_device(self):
        """device mirror of grid%area_t (grid_mod.f90:104-150): dx*dy everywhere, as grid_init fills it
        on a regular grid -- made on first use, a (ny, nx) float64 tensor"""
        if self._area_t_device is None:
            import torch
            if not self.nx:
                raise _cabi.GoceanStop(_cabi.EABORT, "grid%area_t requested before grid_init")
            self._area_t_device = torch.full((self.ny, self.nx), self.dx * self.dy, dtype=torch.float64, device="cuda")
            torch.cuda.current_stream().synchronize()      # in place before a kernel on another stream reads it
        return self._area_t_device

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
    grid.tmask = _make_tmask(grid, tmask)
    grid._tmask_device = None
    grid._area_t_device = None
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


def _make_tmask(grid, tmask):
    """grid_init's T mask (grid_mod.f90:394-455): the grid's own copy of the user's mask on the
    subdomain plus its one-cell ring, rows and columns beyond the ring replicated from it; without a
    user mask the same region is all wet (1).  Cells the reference leaves unset are 0 here.
    Arrays are (ny, nx) row-major = the Fortran (nx, ny)."""
    import numpy as np
    it = grid.subdomain.internal
    xs, xe, ys, ye = it.xstart, it.xstop, it.ystart, it.ystop
    m = np.zeros((grid.ny, grid.nx), dtype=np.int32)
    if tmask is None:
        m[ys - 2:ye + 1, xs - 2:xe + 1] = 1
        return m
    t = np.asarray(tmask)
    m[ys - 2:ye + 1, xs - 2:xe + 1] = t[ys - 2:ye + 1, xs - 2:xe + 1]
    m[ye + 1:, :] = m[ye, :]                   # rows ystop+2 .. ny  <- row ystop+1   (:416-418)
    m[:ys - 2, :] = m[ys - 2, :]               # rows 1 .. ystart-2  <- row ystart-1  (:420-422)
    m[:, :xs - 2] = m[:, xs - 2:xs - 1]        # cols 1 .. xstart-2  <- col xstart-1  (:424-426)
    m[:, xe + 1:] = m[:, xe:xe + 1]            # cols xstop+2 .. nx  <- col xstop+1   (:428-430)
    return m


def halo_plan(grid):
    """one device message plan per grid: every field shares the grid's extents"""
    if grid._halo_plan is None:
        p = C.c_void_p()
        check(_cabi.lib().dlesm_halo_plan_create(C.byref(grid.comm_tables), grid.nx, grid.ny,
                                                 C.byref(p)))
        grid._halo_plan = p
    return grid._halo_plan


def connect_peers(grid, nfields=1):
    """Connect the grid's message plan to its neighbours' mailboxes (dlesm_halo_plan_peer_*): from then on
    dlesm_jacobi5_step_dm[_pipelined] on this grid exchanges by storing straight into the neighbours' memory instead of
    through an RCCL group.  COLLECTIVE: every rank calls it, at the same point.  The blobs travel through the
    torch.distributed group when there is more than one rank (any backend), so no RCCL communicator is needed."""
    L = _cabi.lib()
    plan = halo_plan(grid)
    if L.dlesm_halo_plan_peer_connected(plan):
        return
    rank, n = parallel_mod.get_rank() - 1, parallel_mod.get_num_ranks()
    blob = C.create_string_buffer(_cabi.PEER_BLOB_BYTES)
    check(L.dlesm_halo_plan_peer_export(plan, rank, nfields, blob))
    every = blob.raw
    if n > 1:
        import torch.distributed as dist
        got = [None] * n
        dist.all_gather_object(got, blob.raw)
        every = b"".join(got)
    check(L.dlesm_halo_plan_peer_connect(plan, rank, n, C.create_string_buffer(every, n * _cabi.PEER_BLOB_BYTES)))

