"""Python mirror of field_mod (reference: finite_difference/src/field_mod.f90).

The field's data lives in HBM for its whole life (a torch tensor provides the allocation);
`get_data()/set_data()` are the only host transfers, exactly the two the reference routes
through its device callbacks (field_mod.f90:530-559).
"""
import ctypes as C

import numpy as np

from . import _cabi, grid_mod
from ._cabi import Region, check

GO_U_POINTS, GO_V_POINTS, GO_T_POINTS, GO_F_POINTS, GO_ALL_POINTS = 0, 1, 2, 3, 4
NBOUNDARY = 1


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _cabi.DlesmError(_cabi.ENODEV, "r2d_field needs a GPU: no HIP device is visible "
                                             "(this package has no CPU path)")
    return torch


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def field_bounds(grid, grid_points):
    """set_field_bounds (field_mod.f90:563-624) -> (internal, whole)"""
    internal, whole = Region(), Region()
    check(_cabi.lib().dlesm_field_bounds(grid_points, grid.offset, grid.boundary_conditions[0],
                                         grid.boundary_conditions[1],
                                         C.byref(grid.subdomain.internal), grid.nx, grid.ny,
                                         C.byref(internal), C.byref(whole)))
    return internal, whole


def periodic_halos(grid, grid_points, internal):
    """init_periodic_bc_halos (field_mod.f90:1394-1464): only SW-offset fields on a point type get
    them (the c?_sw_init routines call it; the NE ones stop on periodic boundaries)"""
    if grid_points == GO_ALL_POINTS or grid.offset != grid_mod.GO_OFFSET_SW:
        return []
    src, dst, n = (Region * 4)(), (Region * 4)(), C.c_int()
    check(_cabi.lib().dlesm_periodic_halos(C.byref(internal), grid.boundary_conditions[0],
                                           grid.boundary_conditions[1], src, dst, C.byref(n)))
    out = []
    for k in range(n.value):
        s, d = Region(), Region()
        C.memmove(C.byref(s), C.byref(src[k]), C.sizeof(Region))
        C.memmove(C.byref(d), C.byref(dst[k]), C.sizeof(Region))
        out.append((s, d))
    return out


class r2d_field:
    """r2d_field (field_mod.f90:139-166) with device-resident data"""

    def __init__(self, grid, grid_points, init_global_data=None):
        self.grid = grid
        self.defined_on = grid_points
        self.internal, self.whole = field_bounds(grid, grid_points)
        self.halo = periodic_halos(grid, grid_points, self.internal)   # [(source, dest)], field_mod.f90:1394-1464
        torch = _torch()
        # data(1:grid%nx, 1:grid%ny), zeroed (field_mod.f90:350,375); row-major (ny, nx) here
        self.data = torch.zeros((grid.ny, grid.nx), dtype=torch.float64, device="cuda")
        # The constructor returns a ZEROED field, as the reference's does (allocation and zero fill are synchronous
        # there): the fill above runs on torch's current stream, and a caller that goes on to use the field on another
        # (non-blocking) stream is not ordered behind it -- seen with two ranks sharing a GPU: the fill of a 540 MB
        # field landed after the caller's hash_init.
        torch.cuda.current_stream().synchronize()
        self.data_on_device = True
        self.ntiles = 0
        if init_global_data is not None:                   # field_mod.f90:378-389
            # one strided host-to-device copy of this rank's patch of the global array
            g = np.ascontiguousarray(init_global_data, dtype=np.float64)
            if g.shape != (grid.global_ny, grid.global_nx):
                raise _cabi.DlesmError(_cabi.EINVAL, f"init_global_data has shape {g.shape}, the domain is "
                                                     f"{(grid.global_ny, grid.global_nx)}")
            check(_cabi.lib().dlesm_scatter_inner_f64(g.ctypes.data_as(C.c_void_p), grid.global_nx, grid.global_ny,
                                                      C.byref(grid.subdomain), self.device_ptr, grid.nx, grid.ny))

    # -- raw views ---------------------------------------------------------
    @property
    def device_ptr(self):
        return C.c_void_p(self.data.data_ptr())

    def get_data(self):
        """host copy, shape (ny, nx) (field_mod.f90:530-542): a BLOCKING read, as the reference's read_from_device with
        blocking = .true. -- whatever was enqueued on any stream of this process before the call is in the copy"""
        _torch().cuda.synchronize()
        return self.data.cpu().numpy()

    def set_data(self, array):
        """field_mod.f90:546-559"""
        torch = _torch()
        self.data.copy_(torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)))
        torch.cuda.current_stream().synchronize()          # set_data returns with the data in place (field_mod.f90:546-559)
        return 0

    # -- halo exchange -----------------------------------------------------
    def halo_exchange(self, depth=1, stream=None, dirs=_cabi.DIRS_ALL):
        """halo_exchange (field_mod.f90:1231-1256); depth is ignored as in the reference.  `dirs`:
        the enabled comm directions (bit d-1 for Iplus..Jminus, exchange_generic's comm1..comm4);
        default all four, hence all diagonals, as field_mod.f90:1247-1248"""
        plan = grid_mod.halo_plan(self.grid)
        check(_cabi.lib().dlesm_halo_exchange_f64(plan, self.device_ptr, dirs, _stream_ptr(stream)))

    def gather_inner_data(self):
        """gather_inner_data (field_mod.f90:1313-1390): global (ny, nx) array on rank 1, else None.
        Pack, gather (RCCL) and unpack all run on the device; one copy of the assembled global array
        comes back to the host (dlesm_gather_inner_f64)."""
        from . import parallel_mod
        _torch()
        g = self.grid
        nranks = parallel_mod.get_num_ranks()
        root = parallel_mod.on_master()
        out = np.zeros((g.global_ny, g.global_nx)) if root else None
        check(_cabi.lib().dlesm_gather_inner_f64(self.device_ptr, g.nx, g.ny, C.byref(self.internal),
                                                 C.byref(g.decomp._info), g.decomp.subdomains, nranks,
                                                 out.ctypes.data_as(C.c_void_p) if root else None))
        return out


def field_checksum(field, stream=None):
    """field_checksum (field_mod.f90:1209-1219, 1289-1307): SUM(ABS(internal)) + global_sum"""
    val = C.c_double()
    it = field.internal
    check(_cabi.lib().dlesm_checksum_f64(field.device_ptr, field.grid.nx, field.grid.ny, it.xstart,
                                         it.xstop, it.ystart, it.ystop, C.byref(val),
                                         _stream_ptr(stream)))
    check(_cabi.lib().dlesm_global_sum_f64(C.byref(val)))
    return val.value


def copy_field(field_in, field_out=None, src=None, dest=None, stream=None):
    """copy_field (field_mod.f90:1126-1187): whole-field copy, or patch copy src -> dest
    (Region objects) inside one field."""
    L = _cabi.lib()
    if src is None:
        g = field_in.grid
        check(L.dlesm_copy_patch_f64(field_in.device_ptr, field_out.device_ptr, g.nx, g.ny,
                                     1, 1, 1, 1, g.nx, g.ny, _stream_ptr(stream)))
    else:
        g = field_in.grid
        check(L.dlesm_copy_patch_f64(field_in.device_ptr, field_in.device_ptr, g.nx, g.ny,
                                     src.xstart, src.ystart, dest.xstart, dest.ystart,
                                     src.xstop - src.xstart + 1, src.ystop - src.ystart + 1,
                                     _stream_ptr(stream)))


def set_field(fld, val, stream=None):
    """set_field (field_mod.f90:1191-1202)"""
    g = fld.grid
    check(_cabi.lib().dlesm_fill_f64(fld.device_ptr, g.nx, g.ny, 1, g.nx, 1, g.ny, float(val),
                                     _stream_ptr(stream)))


def free_field(fld):
    """free_field (field_mod.f90:395-403)"""
    fld.data = None
