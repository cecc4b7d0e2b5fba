"""The "PSy layer" for the Python mirror: what PSyclone would generate around each kernel
(`do jj = fld%internal%ystart, ... ; do ji = ...; call kern_code(ji, jj, ...)`, form
infrastructure_mod.f90:32-41) becomes one launch of the matching HIP kernel over the same
index box."""
import ctypes as C

from . import _cabi, grid_mod
from ._cabi import SwParams, check
from .field_mod import _stream_ptr


def invoke_jacobi5(out_fld, in_fld, stream=None):
    """out = 0.25*((w+e)+(s+n)) over out_fld%internal"""
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_stencil5_f64(in_fld.device_ptr, out_fld.device_ptr, g.nx, g.ny,
                                         it.xstart, it.xstop, it.ystart, it.ystop,
                                         _stream_ptr(stream)))


def invoke_stencil9(out_fld, in_fld, coef, stream=None):
    """general 3x3 weighted stencil over out_fld%internal; coef: 9 weights, south-west row first
    (sw, s, se, w, c, e, nw, n, ne) or a 3x3 array indexed [dj+1][di+1]"""
    import numpy as np
    c = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(9))
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_stencil9_f64(in_fld.device_ptr, out_fld.device_ptr,
                                         c.ctypes.data_as(C.POINTER(C.c_double)), g.nx, g.ny,
                                         it.xstart, it.xstop, it.ystart, it.ystop, _stream_ptr(stream)))


def invoke_stencil9_dm(out_fld, in_fld, coef, stream=None):
    """distributed step of a 3x3 weighted kernel: frame, exchange(out) (eight directions when a
    corner weight is non-zero) beside the interior sweep, join"""
    import numpy as np
    c = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(9))
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_stencil9_step_dm(grid_mod.halo_plan(g), in_fld.device_ptr, out_fld.device_ptr,
                                             c.ctypes.data_as(C.POINTER(C.c_double)), g.nx, g.ny,
                                             it.xstart, it.xstop, it.ystart, it.ystop, _stream_ptr(stream)))


def invoke_continuity(ssha, sshn_t, sshn_u, sshn_v, hu, hv, un, vn, rdt, stream=None):
    """the continuity kernel (metadata: GO_GRID_AREA_T): fields on T, U and V points, the grid's cell
    area from the PSy layer (its device mirror), over ssha%internal"""
    g, it = ssha.grid, ssha.internal
    check(_cabi.lib().dlesm_continuity_f64(float(rdt), g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
                                           sshn_t.device_ptr, sshn_u.device_ptr, sshn_v.device_ptr, hu.device_ptr,
                                           hv.device_ptr, un.device_ptr, vn.device_ptr,
                                           C.c_void_p(g.area_t_device.data_ptr()), ssha.device_ptr, _stream_ptr(stream)))


def invoke_jacobi5_masked(out_fld, in_fld, stream=None):
    """the masked Jacobi kernel (metadata: GO_GRID_MASK_T): the PSy layer hands the kernel the
    grid's T mask, here its device mirror"""
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_stencil5_masked_f64(in_fld.device_ptr, out_fld.device_ptr, g.tmask_device_ptr, g.nx,
                                                g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
                                                _stream_ptr(stream)))


def autotune_jacobi5(out_fld, in_fld, stream=None):
    """optional planning call: measure the launch shapes of invoke_jacobi5 for this field geometry
    once (each trial is the same valid step in -> out) and keep the fastest"""
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_stencil5_autotune_f64(in_fld.device_ptr, out_fld.device_ptr, g.nx, g.ny,
                                                  it.xstart, it.xstop, it.ystart, it.ystop,
                                                  _stream_ptr(stream)))


def planned_shape_jacobi5(out_fld):
    """(waves per workgroup, tiles per row, rows per tile, non-temporal stores) for this geometry; the first
    three are 0 before autotune_jacobi5 has run for it"""
    g, it = out_fld.grid, out_fld.internal
    a, b, c, d = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    check(_cabi.lib().dlesm_stencil5_planned_shape(g.nx, it.xstart, it.xstop, it.ystart, it.ystop, C.byref(a), C.byref(b),
                                                   C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


def invoke_jacobi5_x2(out_fld, in_fld, ebox=None, stream=None):
    """TWO Jacobi steps in one sweep: out = J(t) on out_fld%internal, t = J(in) on `ebox`
    (default: the same box, i.e. a fixed boundary ring) and in elsewhere"""
    g, it = out_fld.grid, out_fld.internal
    e = ebox if ebox is not None else it.box()
    check(_cabi.lib().dlesm_stencil5_x2_f64(in_fld.device_ptr, out_fld.device_ptr, g.nx, g.ny,
                                            it.xstart, it.xstop, it.ystart, it.ystop, *e,
                                            _stream_ptr(stream)))


def invoke_jacobi5_multi(out_fld, in_fld, nsteps, ebox=None, grow=(0, 0, 0, 0), stream=None):
    """nsteps (2..8) Jacobi steps in one sweep over out_fld%internal; `ebox` is the last stage
    box (default: the same box, a fixed boundary ring), `grow` the (W, E, S, N) flags"""
    g, it = out_fld.grid, out_fld.internal
    e = ebox if ebox is not None else it.box()
    check(_cabi.lib().dlesm_stencil5_multi_f64(in_fld.device_ptr, out_fld.device_ptr, g.nx, g.ny, nsteps,
                                               it.xstart, it.xstop, it.ystart, it.ystop, *e, *grow,
                                               _stream_ptr(stream)))


def invoke_jacobi5_multi_dm(out_fld, in_fld, nsteps, stream=None):
    """nsteps distributed Jacobi steps with ONE depth-nsteps exchange (hidden behind the interior);
    the grid must have been decomposed with halo_width = nsteps"""
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_jacobi5_multi_step_dm(grid_mod.halo_plan(g), in_fld.device_ptr,
                                                  out_fld.device_ptr, g.nx, g.ny, nsteps, it.xstart,
                                                  it.xstop, it.ystart, it.ystop, _stream_ptr(stream)))


def invoke_jacobi5_dm(out_fld, in_fld, stream=None):
    """distributed step: frame, then exchange(out) hidden behind the interior"""
    g, it = out_fld.grid, out_fld.internal
    plan = grid_mod.halo_plan(g)
    check(_cabi.lib().dlesm_jacobi5_step_dm(plan, in_fld.device_ptr, out_fld.device_ptr, g.nx, g.ny,
                                            it.xstart, it.xstop, it.ystart, it.ystop,
                                            _stream_ptr(stream)))


def invoke_jacobi5_dm_pipelined(out_fld, in_fld, stream=None):
    """the distributed step inside a time loop: returns with the exchange of `out` in flight; the
    next such step waits for it on the device.  Call halo_join(grid) before anything else reads halos."""
    g, it = out_fld.grid, out_fld.internal
    check(_cabi.lib().dlesm_jacobi5_step_dm_pipelined(grid_mod.halo_plan(g), in_fld.device_ptr, out_fld.device_ptr,
                                                      g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
                                                      _stream_ptr(stream)))


def halo_join(grid, stream=None):
    """order `stream` behind the exchange a pipelined step left in flight"""
    check(_cabi.lib().dlesm_halo_plan_join(grid_mod.halo_plan(grid), _stream_ptr(stream)))


def shallow_params(dx, dy, dt):
    """constants of the shallow-water step: tdt = 2*dt (leapfrog)"""
    tdt = dt + dt
    return SwParams(fsdx=4.0 / dx, fsdy=4.0 / dy, tdts8=tdt / 8.0, tdtsdx=tdt / dx, tdtsdy=tdt / dy)


def invoke_shallow_step(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
                                             it.ystart, it.ystop, u.device_ptr, v.device_ptr,
                                             p.device_ptr, uold.device_ptr, vold.device_ptr,
                                             pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                             pnew.device_ptr, _stream_ptr(stream)))


def invoke_shallow_step_sw(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the SW-offset form (the GOcean `shallow` staggering); periodic models follow it with
    apply_periodic_halos on the three new fields"""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_sw_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
                                                it.ystart, it.ystop, u.device_ptr, v.device_ptr,
                                                p.device_ptr, uold.device_ptr, vold.device_ptr,
                                                pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                                pnew.device_ptr, _stream_ptr(stream)))


def apply_periodic_halos(fld, stream=None):
    """the periodic-boundary copies of a field (field_mod.f90:1394-1464), on the device"""
    g = fld.grid
    check(_cabi.lib().dlesm_periodic_halos_apply_f64(fld.device_ptr, g.nx, g.ny, C.byref(fld.internal),
                                                     g.boundary_conditions[0], g.boundary_conditions[1],
                                                     _stream_ptr(stream)))


def apply_periodic_halos_multi(fields, stream=None):
    """the periodic copies of several fields of one grid and internal region in two launches"""
    g = fields[0].grid
    arr = (C.c_void_p * len(fields))(*[f.device_ptr.value for f in fields])
    check(_cabi.lib().dlesm_periodic_halos_apply_multi_f64(arr, len(fields), g.nx, g.ny, C.byref(fields[0].internal),
                                                           g.boundary_conditions[0], g.boundary_conditions[1],
                                                           _stream_ptr(stream)))


def autotune_shallow(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """optional planning call for invoke_shallow_step: time the launch shapes / cache policies once
    for this field geometry (each trial is the same valid step) and keep the fastest"""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_autotune_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
                                                 it.ystart, it.ystop, u.device_ptr, v.device_ptr,
                                                 p.device_ptr, uold.device_ptr, vold.device_ptr,
                                                 pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                                 pnew.device_ptr, _stream_ptr(stream)))


def invoke_shallow_step_dm(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """distributed form: frame of the new fields, one grouped exchange of all three behind the
    interior; unew, vnew, pnew leave with valid halos"""
    g, it = p.grid, p.internal
    plan = grid_mod.halo_plan(g)
    check(_cabi.lib().dlesm_shallow_step_dm(plan, C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
                                            it.ystart, it.ystop, u.device_ptr, v.device_ptr,
                                            p.device_ptr, uold.device_ptr, vold.device_ptr,
                                            pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                            pnew.device_ptr, _stream_ptr(stream)))


def invoke_shallow_step_dm_pipelined(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the distributed shallow-water step inside a time loop: returns with the exchange of the new
    fields in flight; the next such step waits for it on the device.  halo_join(grid) before anything
    else reads their halos."""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_dm_pipelined(grid_mod.halo_plan(g), C.byref(params), g.nx, g.ny,
                                                      it.xstart, it.xstop, it.ystart, it.ystop, u.device_ptr,
                                                      v.device_ptr, p.device_ptr, uold.device_ptr, vold.device_ptr,
                                                      pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                                      pnew.device_ptr, _stream_ptr(stream)))


def halo_exchange_multi(fields, stream=None, dirs=_cabi.DIRS_ALL):
    """halo_exchange(1) of several fields of one grid in a single grouped RCCL launch"""
    g = fields[0].grid
    arr = (C.c_void_p * len(fields))(*[f.device_ptr.value for f in fields])
    check(_cabi.lib().dlesm_halo_exchange_multi_f64(grid_mod.halo_plan(g), arr, len(fields), dirs,
                                                    _stream_ptr(stream)))


def hash_init(fld, seed, box=None, stream=None):
    """synthetic initial condition on `box` (default: the field's whole region), a function of
    the GLOBAL cell index so that every decomposition produces the same global field"""
    g = fld.grid
    b = box or fld.whole
    s = g.subdomain
    gx0 = s.glob.xstart - s.internal.xstart + 1    # global index of local cell 1
    gy0 = s.glob.ystart - s.internal.ystart + 1
    check(_cabi.lib().dlesm_hash_init_f64(fld.device_ptr, g.nx, g.ny, b.xstart, b.xstop, b.ystart,
                                          b.ystop, seed, gx0, gy0, _stream_ptr(stream)))
