"""Independent numpy evaluation of the shallow-water u/v/h update of DESIGN.md section 6.

TEST INFRASTRUCTURE.  The reference (stfc/dl_esm_inf) contains no stencil loop, so the spec of this
update is frozen in DESIGN.md section 6 and nothing in the reference can pin it.  This module is the
second, separately written evaluation that pins `orc_sw_step` (oracle/dlesm_oracle.c): it is NOT a
per-point kernel called from loop nests but whole-array expressions over the complete (ny, ld)
arrays -- every intermediate (cu, cv, z, h) is evaluated wherever its operands exist in the array,
the new time level is then cut out of shifted views.  Same IEEE operations in the same association
order (numpy never contracts a*b+c), hence required to agree with the oracle bit for bit.

Index convention: arrays are (ny, ld) C-order, Fortran element (i, j) = arr[j-1, i-1].
NE offset: u(i,j) is the east face of T(i,j), v(i,j) the north face, z(i,j) the NE corner.
SW offset (sw_step_numpy_sw): u(i,j) is the WEST face of T(i,j), v(i,j) the south face, z(i,j) the SW
corner -- the staggering of the GOcean `shallow` benchmark the SURVEY (section 8 f.2) names.
"""
import numpy as np


def _nan_like(a):
    return np.full_like(a, np.nan)


def sw_intermediates_ne(prm, u, v, p):
    """cu, cv, z, h over the whole array (NaN where an operand would lie outside it)"""
    cu, cv, z, h = _nan_like(p), _nan_like(p), _nan_like(p), _nan_like(p)
    with np.errstate(all="ignore"):                       # padding cells are 0: 0/0 there, never consumed
        # cu(i,j) = 0.5*(p(i+1,j)+p(i,j))*u(i,j)
        cu[:, :-1] = 0.5 * (p[:, 1:] + p[:, :-1]) * u[:, :-1]
        # cv(i,j) = 0.5*(p(i,j+1)+p(i,j))*v(i,j)
        cv[:-1, :] = 0.5 * (p[1:, :] + p[:-1, :]) * v[:-1, :]
        # z(i,j) = (fsdx*(v(i+1,j)-v(i,j)) - fsdy*(u(i,j+1)-u(i,j))) / (p(i,j)+p(i+1,j)+p(i+1,j+1)+p(i,j+1))
        z[:-1, :-1] = (prm.fsdx * (v[:-1, 1:] - v[:-1, :-1]) - prm.fsdy * (u[1:, :-1] - u[:-1, :-1])) / \
                      (p[:-1, :-1] + p[:-1, 1:] + p[1:, 1:] + p[1:, :-1])
        # h(i,j) = p(i,j) + 0.25*(u(i,j)^2 + u(i-1,j)^2 + v(i,j)^2 + v(i,j-1)^2)
        h[1:, 1:] = p[1:, 1:] + 0.25 * (u[1:, 1:] * u[1:, 1:] + u[1:, :-1] * u[1:, :-1] +
                                        v[1:, 1:] * v[1:, 1:] + v[:-1, 1:] * v[:-1, 1:])
    return cu, cv, z, h


def sw_new_level_ne(prm, box, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew):
    """compute_unew / compute_vnew / compute_pnew of the NE staggering from GIVEN cu, cv, z, h arrays (any
    contents): whole-box expressions over shifted views"""
    xs, xe, ys, ye = box

    def S(f, di=0, dj=0):                                 # the box shifted by (di, dj)
        return f[ys - 1 + dj:ye + dj, xs - 1 + di:xe + di]

    # unew = uold + tdts8*(z(i,j)+z(i,j-1))*(cv(i+1,j)+cv(i,j)+cv(i,j-1)+cv(i+1,j-1)) - tdtsdx*(h(i+1,j)-h(i,j))
    if unew is not None:
        S(unew)[...] = S(uold) + prm.tdts8 * (S(z) + S(z, 0, -1)) * \
            (S(cv, 1, 0) + S(cv) + S(cv, 0, -1) + S(cv, 1, -1)) - prm.tdtsdx * (S(h, 1, 0) - S(h))
    # vnew = vold - tdts8*(z(i,j)+z(i-1,j))*(cu(i,j+1)+cu(i-1,j+1)+cu(i-1,j)+cu(i,j)) - tdtsdy*(h(i,j+1)-h(i,j))
    if vnew is not None:
        S(vnew)[...] = S(vold) - prm.tdts8 * (S(z) + S(z, -1, 0)) * \
            (S(cu, 0, 1) + S(cu, -1, 1) + S(cu, -1, 0) + S(cu)) - prm.tdtsdy * (S(h, 0, 1) - S(h))
    # pnew = pold - tdtsdx*(cu(i,j)-cu(i-1,j)) - tdtsdy*(cv(i,j)-cv(i,j-1))
    if pnew is not None:
        S(pnew)[...] = S(pold) - prm.tdtsdx * (S(cu) - S(cu, -1, 0)) - prm.tdtsdy * (S(cv) - S(cv, 0, -1))


def sw_step_numpy(prm, box, u, v, p, uold, vold, pold, unew, vnew, pnew):
    """NE-offset step on the 1-based inclusive box (xs, xe, ys, ye); writes unew/vnew/pnew in place
    on the box only."""
    xs, xe, ys, ye = box
    if xe < xs or ye < ys:
        return
    cu, cv, z, h = sw_intermediates_ne(prm, u, v, p)
    sw_new_level_ne(prm, box, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew)


def sw_intermediates_sw(prm, u, v, p):
    """SW offset (the GOcean `shallow` staggering): cu, cv on the u/v faces WEST/SOUTH of T(i,j),
    z on the SW corner, h on T."""
    cu, cv, z, h = _nan_like(p), _nan_like(p), _nan_like(p), _nan_like(p)
    with np.errstate(all="ignore"):
        # cu(i,j) = 0.5*(p(i,j)+p(i-1,j))*u(i,j)
        cu[:, 1:] = 0.5 * (p[:, 1:] + p[:, :-1]) * u[:, 1:]
        # cv(i,j) = 0.5*(p(i,j)+p(i,j-1))*v(i,j)
        cv[1:, :] = 0.5 * (p[1:, :] + p[:-1, :]) * v[1:, :]
        # z(i,j) = (fsdx*(v(i,j)-v(i-1,j)) - fsdy*(u(i,j)-u(i,j-1))) / (p(i-1,j-1)+p(i,j-1)+p(i,j)+p(i-1,j))
        z[1:, 1:] = (prm.fsdx * (v[1:, 1:] - v[1:, :-1]) - prm.fsdy * (u[1:, 1:] - u[:-1, 1:])) / \
                    (p[:-1, :-1] + p[:-1, 1:] + p[1:, 1:] + p[1:, :-1])
        # h(i,j) = p(i,j) + 0.25*(u(i+1,j)^2 + u(i,j)^2 + v(i,j+1)^2 + v(i,j)^2)
        h[:-1, :-1] = p[:-1, :-1] + 0.25 * (u[:-1, 1:] * u[:-1, 1:] + u[:-1, :-1] * u[:-1, :-1] +
                                            v[1:, :-1] * v[1:, :-1] + v[:-1, :-1] * v[:-1, :-1])
    return cu, cv, z, h


def sw_new_level_sw(prm, box, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew):
    """compute_unew / compute_vnew / compute_pnew of the SW staggering from GIVEN cu, cv, z, h arrays:
    unew = uold + tdts8*(z(i,j+1)+z(i,j))*(cv(i,j+1)+cv(i-1,j+1)+cv(i-1,j)+cv(i,j)) - tdtsdx*(h(i,j)-h(i-1,j))
    vnew = vold - tdts8*(z(i+1,j)+z(i,j))*(cu(i+1,j)+cu(i,j)+cu(i,j-1)+cu(i+1,j-1)) - tdtsdy*(h(i,j)-h(i,j-1))
    pnew = pold - tdtsdx*(cu(i+1,j)-cu(i,j)) - tdtsdy*(cv(i,j+1)-cv(i,j))"""
    xs, xe, ys, ye = box

    def S(f, di=0, dj=0):
        return f[ys - 1 + dj:ye + dj, xs - 1 + di:xe + di]

    if unew is not None:
        S(unew)[...] = S(uold) + prm.tdts8 * (S(z, 0, 1) + S(z)) * \
            (S(cv, 0, 1) + S(cv, -1, 1) + S(cv, -1, 0) + S(cv)) - prm.tdtsdx * (S(h) - S(h, -1, 0))
    if vnew is not None:
        S(vnew)[...] = S(vold) - prm.tdts8 * (S(z, 1, 0) + S(z)) * \
            (S(cu, 1, 0) + S(cu) + S(cu, 0, -1) + S(cu, 1, -1)) - prm.tdtsdy * (S(h) - S(h, 0, -1))
    if pnew is not None:
        S(pnew)[...] = S(pold) - prm.tdtsdx * (S(cu, 1, 0) - S(cu)) - prm.tdtsdy * (S(cv, 0, 1) - S(cv))


def sw_step_numpy_sw(prm, box, u, v, p, uold, vold, pold, unew, vnew, pnew):
    """SW-offset step on the box (the same update with the staggering mirrored)"""
    xs, xe, ys, ye = box
    if xe < xs or ye < ys:
        return
    cu, cv, z, h = sw_intermediates_sw(prm, u, v, p)
    sw_new_level_sw(prm, box, uold, vold, pold, cu, cv, z, h, unew, vnew, pnew)


def time_smooth_numpy(alpha, box, field, field_new, field_old):
    """time_smooth of the GOcean `shallow` leapfrog, in place on the box:
    field_old = field + alpha*(field_new - 2.0*field + field_old)"""
    xs, xe, ys, ye = box
    b = (slice(ys - 1, ye), slice(xs - 1, xe))
    field_old[b] = field[b] + alpha * (field_new[b] - 2.0 * field[b] + field_old[b])


def kernel_numpy(name, sw_offset, prm, box, out, ins, alpha=None):
    """ONE kernel of the GOcean shallow set over the box, by the whole-array expressions above; `ins` in the
    kernel's own argument order: cu(p, u)  cv(p, v)  z(p, u, v)  h(p, u, v)  unew(uold, z, cv, h)
    vnew(vold, z, cu, h)  pnew(pold, cu, cv)  time_smooth(field, field_new, field_old = out)"""
    xs, xe, ys, ye = box
    if xe < xs or ye < ys:
        return
    b = (slice(ys - 1, ye), slice(xs - 1, xe))
    inter = sw_intermediates_sw if sw_offset else sw_intermediates_ne
    new_level = sw_new_level_sw if sw_offset else sw_new_level_ne
    if name in ("cu", "cv", "z", "h"):
        p = ins[0]
        u = ins[1] if name != "cv" else p          # the array a kernel does not take is never consumed by its result
        v = ins[-1] if name != "cu" else p
        out[b] = inter(prm, u, v, p)[("cu", "cv", "z", "h").index(name)][b]
    elif name == "unew":
        uold, z, cv, h = ins
        new_level(prm, box, uold, None, None, None, cv, z, h, out, None, None)
    elif name == "vnew":
        vold, z, cu, h = ins
        new_level(prm, box, None, vold, None, cu, None, z, h, None, out, None)
    elif name == "pnew":
        pold, cu, cv = ins
        new_level(prm, box, None, None, pold, cu, cv, None, None, None, None, out)
    elif name == "time_smooth":
        assert ins[2] is out
        time_smooth_numpy(alpha, box, ins[0], ins[1], out)
    else:
        raise ValueError(name)


def kernel_scalars(name, prm, alpha=0.001):
    """the two real scalars the kernel's launch entry takes"""
    return {"z": (prm.fsdx, prm.fsdy), "unew": (prm.tdts8, prm.tdtsdx), "vnew": (prm.tdts8, prm.tdtsdy),
            "pnew": (prm.tdtsdx, prm.tdtsdy), "time_smooth": (alpha, 0.0)}.get(name, (0.0, 0.0))


# cells next to the box that a kernel reads (W, E, S, N), per staggering: the box must leave that ring free
KERNEL_RING = {
    False: {"cu": (0, 1, 0, 0), "cv": (0, 0, 0, 1), "z": (0, 1, 0, 1), "h": (1, 0, 1, 0), "unew": (0, 1, 1, 0),
            "vnew": (1, 0, 0, 1), "pnew": (1, 0, 1, 0), "time_smooth": (0, 0, 0, 0)},
    True: {"cu": (1, 0, 0, 0), "cv": (0, 0, 1, 0), "z": (1, 0, 1, 0), "h": (0, 1, 0, 1), "unew": (1, 0, 0, 1),
           "vnew": (0, 1, 1, 0), "pnew": (0, 1, 0, 1), "time_smooth": (0, 0, 0, 0)},
}
KERNEL_NIN = {"cu": 2, "cv": 2, "z": 3, "h": 3, "unew": 4, "vnew": 4, "pnew": 3, "time_smooth": 3}


class Params:
    """fsdx = 4/dx, fsdy = 4/dy, tdt = 2 dt, tdts8 = tdt/8, tdtsdx = tdt/dx, tdtsdy = tdt/dy"""

    def __init__(self, dx, dy, dt):
        tdt = dt + dt
        self.fsdx, self.fsdy = 4.0 / dx, 4.0 / dy
        self.tdts8, self.tdtsdx, self.tdtsdy = tdt / 8.0, tdt / dx, tdt / dy


# ------------------------------------------------------------------------------------------------
# The multi-step protocol of tests/golden/sw_numpy_64x48.json (generator: make_sw_golden.py):
# hash initial state, three time levels that share one boundary ring, leapfrog by buffer rotation.
def initial_state(hash_field, seed, ny_arr, ld, whole):
    """u, v in [-0.5, 0.5), p in [1, 2) from the counter hash on the `whole` region (local cell 1 =
    global cell 0), the shift applied to the complete array like the device tests do; old and new time
    levels start as copies, so all three share the same fixed boundary ring."""
    xlo, xhi, ylo, yhi = whole
    cur = []
    for k, shift in enumerate((-0.5, -0.5, 1.0)):
        f = hash_field(seed + k, ny_arr, ld, 0, 0, xlo, xhi, ylo, yhi)
        cur.append(f + shift)
    old = [f.copy() for f in cur]
    new = [f.copy() for f in cur]
    return cur, old, new


def leapfrog(step, nsteps, cur, old, new, on_step=None):
    """step(cur(u,v,p), old(u,v,p), new(u,v,p)) writes `new` on the box; then rotate."""
    for k in range(1, nsteps + 1):
        step(cur, old, new)
        old, cur, new = cur, new, old
        if on_step:
            on_step(k, cur)
    return cur, old, new


def digest(field, box):
    """sha256 over the bytes of the box of a (ny, ld) array, row by row: pins every bit"""
    import hashlib
    xs, xe, ys, ye = box
    return hashlib.sha256(np.ascontiguousarray(field[ys - 1:ye, xs - 1:xe]).tobytes()).hexdigest()


def abs_sum(field, box):
    """exactly rounded SUM(ABS()) of the box (math.fsum)"""
    import math
    xs, xe, ys, ye = box
    return math.fsum(np.abs(field[ys - 1:ye, xs - 1:xe]).ravel().tolist())
