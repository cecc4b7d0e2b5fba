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


def halo_connect_peers(grid, nfields=1):
    """collective: connect the grid's plan to the neighbours' mailboxes (grid_mod.connect_peers) -- the distributed
    Jacobi steps then exchange with stores over xGMI instead of an RCCL group"""
    grid_mod.connect_peers(grid, nfields)


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


def invoke_shallow_step_x2(params, u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2, stream=None):
    """TWO leapfrog steps in one launch (dlesm_shallow_step_x2_f64): level n+1 into unew / vnew / pnew, level n+2 into
    unew2 / vnew2 / pnew2 -- the bits of two invoke_shallow_step calls at 48 instead of 72 B/cell/step.  Time loop:
    (cur, old, new1, new2) <- (new2, new1, old, cur) after every call."""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_x2_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
                                                *[f.device_ptr for f in (u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2)],
                                                _stream_ptr(stream)))


def invoke_shallow_step_smooth_x2(params, alpha, u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2, stream=None):
    """TWO filtered leapfrog steps (update + time_smooth, twice) in one launch (dlesm_shallow_step_smooth_x2_f64): level n+2 into
    unew2.., the filtered level n+1 into uold2..; inputs untouched.  Time loop: ping-pong (cur, old) <-> (unew2.., uold2..)."""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_smooth_x2_f64(
        C.byref(params), alpha, g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
        *[f.device_ptr for f in (u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2)], _stream_ptr(stream)))


def invoke_shallow_step_sw(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the SW-offset form (the GOcean `shallow` staggering); periodic models follow it with
    apply_periodic_halos on the three new fields"""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_sw_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
                                                it.ystart, it.ystop, u.device_ptr, v.device_ptr,
                                                p.device_ptr, uold.device_ptr, vold.device_ptr,
                                                pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                                pnew.device_ptr, _stream_ptr(stream)))


# ---- the GOcean `shallow` kernels one by one: what an unmodified generated PSy layer calls ----------------
def _kernel_box(out_fld, box):
    """the loop bounds of the nest: the written field's internal region (ITERATES_OVER = GO_INTERNAL_PTS,
    kernel_mod.f90:28-50) unless the caller's PSy layer runs the kernel over another box"""
    return tuple(box) if box is not None else out_fld.internal.box()


def invoke_compute_cu(cu, p, u, box=None, stream=None):
    """`call compute_cu_code(ji, jj, cu%data, p%data, u%data)` over cu%internal"""
    g = cu.grid
    check(_cabi.lib().dlesm_compute_cu_f64(g.offset, g.nx, g.ny, *_kernel_box(cu, box), cu.device_ptr, p.device_ptr,
                                           u.device_ptr, _stream_ptr(stream)))


def invoke_compute_cv(cv, p, v, box=None, stream=None):
    g = cv.grid
    check(_cabi.lib().dlesm_compute_cv_f64(g.offset, g.nx, g.ny, *_kernel_box(cv, box), cv.device_ptr, p.device_ptr,
                                           v.device_ptr, _stream_ptr(stream)))


def invoke_compute_z(z, p, u, v, box=None, stream=None):
    """fsdx = 4/dx, fsdy = 4/dy from the grid (the kernel's GO_GRID_DX_CONST / GO_GRID_DY_CONST arguments)"""
    g = z.grid
    check(_cabi.lib().dlesm_compute_z_f64(g.offset, g.nx, g.ny, *_kernel_box(z, box), 4.0 / g.dx, 4.0 / g.dy,
                                          z.device_ptr, p.device_ptr, u.device_ptr, v.device_ptr, _stream_ptr(stream)))


def invoke_compute_h(h, p, u, v, box=None, stream=None):
    g = h.grid
    check(_cabi.lib().dlesm_compute_h_f64(g.offset, g.nx, g.ny, *_kernel_box(h, box), h.device_ptr, p.device_ptr,
                                          u.device_ptr, v.device_ptr, _stream_ptr(stream)))


def invoke_compute_unew(unew, uold, z, cv, h, tdt, box=None, stream=None):
    """tdt = 2*dt in a leapfrog step; tdts8 = tdt/8, tdtsdx = tdt/dx"""
    g = unew.grid
    check(_cabi.lib().dlesm_compute_unew_f64(g.offset, g.nx, g.ny, *_kernel_box(unew, box), tdt / 8.0, tdt / g.dx,
                                             unew.device_ptr, uold.device_ptr, z.device_ptr, cv.device_ptr,
                                             h.device_ptr, _stream_ptr(stream)))


def invoke_compute_vnew(vnew, vold, z, cu, h, tdt, box=None, stream=None):
    g = vnew.grid
    check(_cabi.lib().dlesm_compute_vnew_f64(g.offset, g.nx, g.ny, *_kernel_box(vnew, box), tdt / 8.0, tdt / g.dy,
                                             vnew.device_ptr, vold.device_ptr, z.device_ptr, cu.device_ptr,
                                             h.device_ptr, _stream_ptr(stream)))


def invoke_compute_pnew(pnew, pold, cu, cv, tdt, box=None, stream=None):
    g = pnew.grid
    check(_cabi.lib().dlesm_compute_pnew_f64(g.offset, g.nx, g.ny, *_kernel_box(pnew, box), tdt / g.dx, tdt / g.dy,
                                             pnew.device_ptr, pold.device_ptr, cu.device_ptr, cv.device_ptr,
                                             _stream_ptr(stream)))


def invoke_time_smooth(field, field_new, field_old, alpha, box=None, stream=None):
    """field_old = field + alpha*(field_new - 2*field + field_old) over field_old%internal"""
    g = field_old.grid
    check(_cabi.lib().dlesm_time_smooth_f64(g.nx, g.ny, *_kernel_box(field_old, box), float(alpha), field.device_ptr,
                                            field_new.device_ptr, field_old.device_ptr, _stream_ptr(stream)))


def invoke_shallow_kernel_sequence(tdt, u, v, p, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew, stream=None):
    """One time step the way a generated PSy layer runs it: seven loop nests, every intermediate through HBM
    (224 B/cell).  Non-periodic grids: cu, cv, z, h over the internal region grown towards their consumers (all
    operands stay inside the boundary ring); periodic (SW-offset) grids: over the internal region, followed by
    their periodic copies, as the benchmark does.  Bit-identical to invoke_shallow_step / invoke_shallow_step_sw."""
    g = p.grid
    xs, xe, ys, ye = p.internal.box()
    periodic = GO_BC_PERIODIC_ in g.boundary_conditions[:2]
    if periodic:
        grown = dict(cu=None, cv=None, z=None, h=None)
    elif g.offset == grid_mod.GO_OFFSET_NE:
        grown = dict(cu=(xs - 1, xe, ys, ye + 1), cv=(xs, xe + 1, ys - 1, ye), z=(xs - 1, xe, ys - 1, ye),
                     h=(xs, xe + 1, ys, ye + 1))
    else:
        grown = dict(cu=(xs, xe + 1, ys - 1, ye), cv=(xs - 1, xe, ys, ye + 1), z=(xs, xe + 1, ys, ye + 1),
                     h=(xs - 1, xe, ys - 1, ye))
    invoke_compute_cu(cu, p, u, grown["cu"], stream)
    invoke_compute_cv(cv, p, v, grown["cv"], stream)
    invoke_compute_z(z, p, u, v, grown["z"], stream)
    invoke_compute_h(h, p, u, v, grown["h"], stream)
    if periodic:
        apply_periodic_halos_multi([cu, cv, z, h], stream)
    invoke_compute_unew(unew, uold, z, cv, h, tdt, None, stream)
    invoke_compute_vnew(vnew, vold, z, cu, h, tdt, None, stream)
    invoke_compute_pnew(pnew, pold, cu, cv, tdt, None, stream)


GO_BC_PERIODIC_ = grid_mod.GO_BC_PERIODIC


def invoke_shallow_step_sw_periodic(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the SW-offset step over the internal region AND the periodic copies of the three new fields in one launch
    (== invoke_shallow_step_sw + apply_periodic_halos_multi, bit for bit)"""
    g = p.grid
    check(_cabi.lib().dlesm_shallow_step_sw_periodic_f64(C.byref(params), g.nx, g.ny, C.byref(p.internal),
                                                         g.boundary_conditions[0], g.boundary_conditions[1],
                                                         u.device_ptr, v.device_ptr, p.device_ptr, uold.device_ptr,
                                                         vold.device_ptr, pold.device_ptr, unew.device_ptr,
                                                         vnew.device_ptr, pnew.device_ptr, _stream_ptr(stream)))


def invoke_shallow_step_sw_x2_periodic(params, u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2, stream=None):
    """two steps of the SW-offset periodic model in one launch (== two invoke_shallow_step_sw_periodic calls)"""
    g = p.grid
    check(_cabi.lib().dlesm_shallow_step_sw_x2_periodic_f64(
        C.byref(params), g.nx, g.ny, C.byref(p.internal), g.boundary_conditions[0], g.boundary_conditions[1],
        *[f.device_ptr for f in (u, v, p, uold, vold, pold, unew, vnew, pnew, unew2, vnew2, pnew2)], _stream_ptr(stream)))


def invoke_shallow_step_sw_smooth_x2_periodic(params, alpha, u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2, stream=None):
    """two WHOLE time steps of the GOcean `shallow` benchmark (update + time_smooth + periodic images, twice) in one launch:
    level n+2 into unew2.., the filtered level n+1 into uold2..; ping-pong between the two sextets"""
    g = p.grid
    check(_cabi.lib().dlesm_shallow_step_sw_smooth_x2_periodic_f64(
        C.byref(params), float(alpha), g.nx, g.ny, C.byref(p.internal), g.boundary_conditions[0], g.boundary_conditions[1],
        *[f.device_ptr for f in (u, v, p, uold, vold, pold, unew2, vnew2, pnew2, uold2, vold2, pold2)], _stream_ptr(stream)))


def invoke_shallow_step_smooth(params, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """one whole leapfrog step of the GOcean benchmark in one launch (NE offset): the u/v/h update + time_smooth of the
    old level in place (== invoke_shallow_step + 3 x invoke_time_smooth, bit for bit).  Afterwards rotate u <- unew."""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_step_smooth_f64(C.byref(params), float(alpha), g.nx, g.ny, it.xstart, it.xstop, it.ystart,
                                                    it.ystop, u.device_ptr, v.device_ptr, p.device_ptr, uold.device_ptr,
                                                    vold.device_ptr, pold.device_ptr, unew.device_ptr, vnew.device_ptr,
                                                    pnew.device_ptr, _stream_ptr(stream)))


def invoke_shallow_step_sw_smooth_periodic(params, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the same for the SW-offset periodic model: step + time_smooth + the periodic images of the new and of the filtered
    old level, one launch"""
    g = p.grid
    check(_cabi.lib().dlesm_shallow_step_sw_smooth_periodic_f64(C.byref(params), float(alpha), g.nx, g.ny, C.byref(p.internal),
                                                                g.boundary_conditions[0], g.boundary_conditions[1],
                                                                u.device_ptr, v.device_ptr, p.device_ptr, uold.device_ptr,
                                                                vold.device_ptr, pold.device_ptr, unew.device_ptr,
                                                                vnew.device_ptr, pnew.device_ptr, _stream_ptr(stream)))


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


def autotune_shallow_sw(params, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None):
    """the planning call of the SW-offset step (invoke_shallow_step_sw / invoke_shallow_step_sw_periodic)"""
    g, it = p.grid, p.internal
    check(_cabi.lib().dlesm_shallow_autotune_sw_f64(C.byref(params), g.nx, g.ny, it.xstart, it.xstop,
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


def invoke_shallow_step_smooth_dm(params, alpha, u, v, p, uold, vold, pold, unew, vnew, pnew, stream=None, pipelined=False):
    """the distributed step with the Asselin filter of the old level folded in (joined or time-loop form)"""
    g, it = p.grid, p.internal
    fn = _cabi.lib().dlesm_shallow_step_smooth_dm_pipelined if pipelined else _cabi.lib().dlesm_shallow_step_smooth_dm
    check(fn(grid_mod.halo_plan(g), C.byref(params), float(alpha), g.nx, g.ny, it.xstart, it.xstop, it.ystart, it.ystop,
             u.device_ptr, v.device_ptr, p.device_ptr, uold.device_ptr, vold.device_ptr, pold.device_ptr, unew.device_ptr,
             vnew.device_ptr, pnew.device_ptr, _stream_ptr(stream)))


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
