"""Known-answer scenarios restated from the reference's own tests, shared by the
oracle tests and the product (C-ABI / HIP) tests.  Pure numpy, no reference
source is read at run time.

  * hill() / init_field_hill / check_hill_halos : tests/dist_mem/test_halos.f90:142-265
  * gsum fill                                    : tests/dist_mem/test_gsum.f90:138-154
  * unique_global_value                          : tests/dist_mem/test_reduction.f90:118-123
"""
import numpy as np

GO_U, GO_V, GO_T, GO_F, GO_ALL = 0, 1, 2, 3, 4
OFFSET_SW, OFFSET_NE = 0, 3
BC_PERIODIC, BC_EXTERNAL, BC_NONE = 0, 1, 2

# the reference's dist_mem test matrix (tests/dist_mem/Makefile:64-80): (jpiglo, jpjglo, nranks)
HALO_CASES = [(10, 4, 2), (4, 10, 2), (10, 10, 4), (10, 10, 6)]
GSUM_CASES = [(4, 10, 4), (4, 10, 6)]
REDUCTION_CASES = [(10, 10, 4), (10, 10, 6)]


def hill_value(ptype, xt, yt, offset=OFFSET_NE, dx=1.0, dy=1.0):
    """test_halos.f90:153-189 : real(10000.0*xpos + ypos) -- note the default-real
    (single precision) rounding of the reference expression."""
    xpos, ypos = float(xt), float(yt)
    s = 0.5 if offset == OFFSET_NE else -0.5
    if ptype == GO_U:
        xpos += s * dx
    elif ptype == GO_F:
        xpos += s * dx
        ypos += s * dy
    elif ptype == GO_V:
        ypos += s * dy
    return float(np.float32(10000.0 * xpos + ypos))


def t_coords(gxstart, gystart, ixstart, iystart, nx, ny, dx=1.0, dy=1.0):
    """grid%xt / grid%yt as grid_mod.f90:536-556 fills them: the first internal
    column carries global%xstart*dx, everything else is +-dx from it."""
    ji = np.arange(1, nx + 1)
    jj = np.arange(1, ny + 1)
    xt = (gxstart + (ji - ixstart)) * dx
    yt = (gystart + (jj - iystart)) * dy
    return xt, yt


def init_field_hill(ptype, nx, ny, internal, gxstart, gystart):
    """test_halos.f90:127-151.  internal = (xstart, xstop, ystart, ystop), 1-based.
    Returns the (ny, nx) array (row index = j)."""
    xs, xe, ys, ye = internal
    xt, yt = t_coords(gxstart, gystart, xs, ys, nx, ny)
    f = np.zeros((ny, nx))
    for jj in range(ys, ye + 1):
        for ji in range(xs, xe + 1):
            f[jj - 1, ji - 1] = hill_value(ptype, xt[ji - 1], yt[jj - 1])
    # externals: plausible but wrong (replicate the nearest internal value)
    f[:, :xs - 1] = f[:, xs - 1:xs]
    f[:, xe:] = f[:, xe - 1:xe]
    f[:ys - 1, :] = f[ys - 1:ys, :]
    f[ye:, :] = f[ye - 1:ye, :]
    return f


def check_hill_halos(f, ptype, internal, sub_global, global_nx, global_ny, corners=False):
    """test_halos.f90:191-265 : every depth-1 halo cell that faces a neighbour must equal
    hill() of its own (global) position.  sub_global = (gxstart, gxstop, gystart, gystop).
    Returns the list of mismatching (ji, jj, got, want).  With corners=True the diagonal
    halo cells are checked too (the reference test does not, the 9-point stencil needs them)."""
    xs, xe, ys, ye = internal
    gxs, gxe, gys, gye = sub_global
    ny, nx = f.shape
    xt, yt = t_coords(gxs, gys, xs, ys, nx, ny)
    bad = []

    def chk(ji, jj):
        want = hill_value(ptype, xt[ji - 1], yt[jj - 1])
        got = f[jj - 1, ji - 1]
        if abs(got - want) > 1.0e-8:
            bad.append((ji, jj, got, want))

    has_w, has_e = gxs > 1, gxe < global_nx
    has_s, has_n = gys > 1, gye < global_ny
    if has_w:
        for jj in range(ys, ye + 1):
            chk(xs - 1, jj)
    if has_e:
        for jj in range(ys, ye + 1):
            chk(xe + 1, jj)
    if has_s:
        for ji in range(xs, xe + 1):
            chk(ji, ys - 1)
    if has_n:
        for ji in range(xs, xe + 1):
            chk(ji, ye + 1)
    if corners:
        if has_w and has_s: chk(xs - 1, ys - 1)
        if has_w and has_n: chk(xs - 1, ye + 1)
        if has_e and has_s: chk(xe + 1, ys - 1)
        if has_e and has_n: chk(xe + 1, ye + 1)
    return bad


def gsum_field(nx, ny, internal):
    """test_gsum.f90:138-154 : internal = 1.0, everything else -100.0"""
    xs, xe, ys, ye = internal
    f = np.full((ny, nx), -100.0)
    f[ys - 1:ye, xs - 1:xe] = 1.0
    return f


def unique_global(gnx, gny):
    """test_reduction.f90:118-123 : (i-1) + (j-1)*n as a (gny, gnx) array"""
    i = np.arange(gnx)[None, :]
    j = np.arange(gny)[:, None]
    return (i + j * gnx).astype(np.float64)
