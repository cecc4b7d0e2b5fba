"""GPU tests (-m gpu): two leapfrog steps of the shallow-water update per launch (dlesm_shallow_step_x2_f64, DESIGN.md
section 5.4) against the ORACLE's step applied twice with the leapfrog rotation -- every bit of both new levels, ring cells
included -- on shapes that exercise the rim handling (tiles on every edge of the box, one-tile boxes, boxes narrower than a wave
tile, odd leading dimensions, heights that are not a multiple of the tile height), the forced fall-back (two single steps), time
loops of several double steps against single steps, and the full-size configuration against the single-step kernel."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 20261004 + 40
NAMES = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew", "unew2", "vnew2", "pnew2"]


@pytest.fixture(scope="module")
def D():
    import torch
    import dl_esm_inf_amd as d
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    d.parallel_init(0, 1)
    return d


def _grid(D, nx, ny, alignment):
    import os
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    return g


def _fields(D, g, seed=SEED):
    """twelve fields; levels n and n-1 are hash data in physically sane ranges over the WHOLE array; the boundary ring of every
    level holds the same fixed values (what a non-periodic model keeps there): the ring of u / v / p"""
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in NAMES}
    for k, n in enumerate(NAMES[:6]):
        D.psy.hash_init(F[n], seed + k)
        F[n].data.mul_(0.1)
        F[n].data.add_(1.0 if n[0] == "p" else -0.05)
    for k, n in enumerate(NAMES[6:]):
        D.copy_field(F[NAMES[k % 3]], F[n])          # ring = the ring of level n; the interior is overwritten by the step
    it = F["p"].internal
    for k, n in enumerate(NAMES[3:6]):               # ... and level n-1 has that ring too (its interior stays its own)
        keep = F[n].data[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop].clone()
        D.copy_field(F[NAMES[k]], F[n])
        F[n].data[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop] = keep
    return F


def _oracle_two_steps(prm, g, box, H):
    """level n+1 and n+2 by two oracle steps (the second reads the first's output incl. its untouched ring)"""
    n1 = [H[n].copy() for n in NAMES[6:9]]
    n2 = [H[n].copy() for n in NAMES[9:]]
    O.sw_step(prm, g.nx, box, H["u"], H["v"], H["p"], H["uold"], H["vold"], H["pold"], *n1)
    O.sw_step(prm, g.nx, box, *n1, H["u"], H["v"], H["p"], *n2)
    return n1, n2


CASES = [(5, 4, None), (5, 4, 2), (1, 1, 2), (2, 1, 2), (1, 3, 2), (64, 48, 8), (123, 3, 2), (124, 5, 2), (125, 2, 2), (126, 7, 64),
         (127, 9, None), (257, 129, None), (300, 70, 64), (1000, 37, 64), (2100, 33, 64), (4000, 9, 2), (8000, 9, 64)]


TUNES = [dict(), dict(sw_x2_stack=0), dict(sw_x2_rows=2, sw_x2_nt=7), dict(sw_x2_rows=6, sw_x2_nt=6, sw_x2_stack=8), dict(sw_x2_nt=3, sw_x2_stack=2, sw_x2_pad=3),
         dict(sw_x2_rows=2, sw_x2_nt=0, sw_x2_stack=0)]      # (all but the first: comparison forms, libdlesm_hip_lab.so)
X2_DEFAULTS = dict(sw_x2_rows=0, sw_x2_nt=2, sw_x2_stack=4, sw_x2_pad=0)


@pytest.mark.parametrize("tune", TUNES, ids=lambda t: "-".join(f"{k[6:]}{v}" for k, v in t.items()) or "default")
@pytest.mark.parametrize("nx,ny,alignment", CASES)
def test_two_steps_per_launch_match_the_oracle(D, nx, ny, alignment, tune):
    import torch
    L = D._cabi.lib()
    g = _grid(D, nx, ny, alignment)
    F = _fields(D, g)
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    H = {n: F[n].get_data() for n in NAMES}
    box = F["p"].internal.box()
    n1, n2 = _oracle_two_steps(prm, g, box, H)
    for k, v in tune.items():
        L.dlesm_set_tuning(k.encode(), v)
    try:
        D.psy.invoke_shallow_step_x2(prm, *[F[n] for n in NAMES])
        torch.cuda.synchronize()
    finally:
        for k in tune:
            L.dlesm_set_tuning(k.encode(), X2_DEFAULTS[k])
    for name, want in zip(NAMES[6:], n1 + n2):
        got = F[name].get_data()
        assert np.array_equal(got, want), (name, int(np.count_nonzero(got != want)), np.argwhere(got != want)[:4])
    for n in NAMES[:6]:                              # the inputs are untouched
        assert np.array_equal(F[n].get_data(), H[n]), n


@pytest.mark.parametrize("nx,ny,alignment", [(300, 70, 64), (127, 9, None)])
def test_forced_fallback_is_two_single_steps(D, nx, ny, alignment):
    import torch
    L = D._cabi.lib()
    g = _grid(D, nx, ny, alignment)
    F = _fields(D, g, SEED + 100)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 60.0)
    H = {n: F[n].get_data() for n in NAMES}
    n1, n2 = _oracle_two_steps(prm, g, F["p"].internal.box(), H)
    for key in (b"sw_x2_fused", b"sw_kernel"):
        for n in NAMES[6:]:
            F[n].set_data(H[n])
        L.dlesm_set_tuning(key, 0 if key == b"sw_x2_fused" else 1)
        try:
            D.psy.invoke_shallow_step_x2(prm, *[F[n] for n in NAMES])
            torch.cuda.synchronize()
        finally:
            L.dlesm_set_tuning(key, 1 if key == b"sw_x2_fused" else 0)
        for name, want in zip(NAMES[6:], n1 + n2):
            assert np.array_equal(F[name].get_data(), want), (key, name)


def test_refusals(D):
    L = D._cabi.lib()
    g = _grid(D, 64, 48, 8)
    F = _fields(D, g)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 60.0)
    it = F["p"].internal
    ptrs = [F[n].device_ptr for n in NAMES]
    call = lambda p_, box: L.dlesm_shallow_step_x2_f64(C.byref(prm), g.nx, g.ny, *box, *p_, None)      # noqa: E731
    assert call(ptrs, it.box()) == 0
    aliased = list(ptrs)
    aliased[9] = aliased[3]                           # level n+2 into the arrays of level n-1: refused (not in place)
    assert call(aliased, it.box()) == D._cabi.EINVAL and b"overlap" in L.dlesm_last_error()
    shifted = list(ptrs)
    shifted[10] = C.c_void_p(ptrs[4].value + 16 * g.nx)     # vnew2 two rows into vold: a shifted alias
    assert call(shifted, it.box()) == D._cabi.EINVAL and b"overlap" in L.dlesm_last_error()
    assert call(ptrs, (1, it.xstop, it.ystart, it.ystop)) == D._cabi.EINVAL          # no room for the stencil ring
    assert call(ptrs, (5, 4, it.ystart, it.ystop)) == 0                              # an empty box: a zero-trip loop nest


@pytest.mark.parametrize("nx,ny,alignment,pairs", [(257, 129, None, 3), (640, 200, 64, 4)])
def test_time_loop_of_double_steps_equals_single_steps(D, nx, ny, alignment, pairs):
    """2 x pairs leapfrog steps: double steps with the four-level rotation against single steps with the three-level one"""
    import torch
    g = _grid(D, nx, ny, alignment)
    A = _fields(D, g, SEED + 7)
    B = _fields(D, g, SEED + 7)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 30.0)
    cur, old, n1, n2 = ([A[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6, 9))
    for _ in range(pairs):
        D.psy.invoke_shallow_step_x2(prm, *cur, *old, *n1, *n2)
        cur, old, n1, n2 = n2, n1, old, cur
    c, o, n = ([B[q] for q in NAMES[k:k + 3]] for k in (0, 3, 6))
    for _ in range(2 * pairs):
        D.psy.invoke_shallow_step(prm, *c, *o, *n)
        c, o, n = n, c, o
    torch.cuda.synchronize()
    for x, y in zip(cur + old, c + o):
        assert torch.equal(x.data, y.data)


def test_full_size_equals_two_single_steps(D):
    """BASELINE configs[3]'s size: 8192^2, alignment 64, both new levels against the single-step kernel"""
    import torch
    g = _grid(D, 8192, 8192, 64)
    F = _fields(D, g, SEED + 9)
    chk = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(6)]
    for x, n in zip(chk, NAMES[6:]):
        D.copy_field(F[n], x)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    D.psy.invoke_shallow_step_x2(prm, *[F[n] for n in NAMES])
    D.psy.invoke_shallow_step(prm, *[F[n] for n in NAMES[:6]], *chk[:3])
    D.psy.invoke_shallow_step(prm, *chk[:3], *[F[n] for n in NAMES[:3]], *chk[3:])
    torch.cuda.synchronize()
    for x, n in zip(chk, NAMES[6:]):
        assert torch.equal(x.data, F[n].data), n


# ---- with the Asselin filter after each step: two whole time steps of the GOcean loop per launch --------------------------------
@pytest.mark.parametrize("nx,ny,alignment", [(5, 4, None), (64, 48, 8), (127, 9, None), (257, 129, None), (300, 70, 64), (1000, 37, 64), (2100, 33, 64)])
@pytest.mark.parametrize("fallback,x2_rows", [(False, 0), (True, 0), (False, 2), (False, 4)])   # (rows 2 / 4: the lab build's tile heights)
def test_two_filtered_steps_per_launch(D, nx, ny, alignment, fallback, x2_rows):
    """dlesm_shallow_step_smooth_x2_f64 against the one-launch filtered step applied twice with the loop's rotation (itself checked
    against the oracle's loop nests in tests/test_gpu_shallow_kernels.py), and a longer ping-pong loop against that loop"""
    import torch
    L = D._cabi.lib()
    g = _grid(D, nx, ny, alignment)
    alpha = 0.001
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    A, B = _fields(D, g, SEED + 21), _fields(D, g, SEED + 21)
    a_cur, a_old, a_n2, a_o2 = ([A[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6, 9))
    for src, dst in zip(a_cur, a_o2):                 # every array of the run carries the same boundary ring
        D.copy_field(src, dst)
    b_cur, b_old, b_new = ([B[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6))
    L.dlesm_set_tuning(b"sw_x2_fused", 0 if fallback else 1)
    if x2_rows:
        L.dlesm_set_tuning(b"sw_x2_rows", x2_rows)
    try:
        for pair in range(3):
            keep = [f.data.clone() for f in a_cur + a_old]
            D.psy.invoke_shallow_step_smooth_x2(prm, alpha, *a_cur, *a_old, *a_n2, *a_o2)
            torch.cuda.synchronize()
            assert all(torch.equal(k, f.data) for k, f in zip(keep, a_cur + a_old)), "inputs modified"
            a_cur, a_old, a_n2, a_o2 = a_n2, a_o2, a_cur, a_old
            for _ in range(2):
                D.psy.invoke_shallow_step_smooth(prm, alpha, *b_cur, *b_old, *b_new)
                b_cur, b_old, b_new = b_new, b_old, b_cur          # uold already holds the filtered former current
            torch.cuda.synchronize()
            it = A["p"].internal
            cut = lambda f: f.data[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop]      # noqa: E731
            for x, y in zip(a_cur + a_old, b_cur + b_old):
                assert torch.equal(cut(x), cut(y)), (pair, int((cut(x) != cut(y)).sum()))
    finally:
        L.dlesm_set_tuning(b"sw_x2_fused", 1)
        if x2_rows:
            L.dlesm_set_tuning(b"sw_x2_rows", 0)


# ---- the SW-offset, doubly periodic model (the GOcean `shallow` benchmark's configuration) --------------------------------------
def _grid_sw(D, nx, ny, alignment):
    import os
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_PERIODIC, D.GO_BC_PERIODIC, D.GO_BC_NONE), D.GO_OFFSET_SW)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    return g


def _periodic_fields(D, g, seed):
    """twelve fields of the periodic model: levels n and n-1 hash data on the internal region with their periodic images; the
    output arrays start as garbage (-3) everywhere, halos included: whatever is valid afterwards the launch has written"""
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in NAMES}
    it = F["p"].internal
    for k, n in enumerate(NAMES[:6]):
        D.psy.hash_init(F[n], seed + k, box=it)
        F[n].data.mul_(0.1)
        F[n].data.add_(1.0 if n[0] == "p" else -0.05)
        D.psy.apply_periodic_halos(F[n])
    for n in NAMES[6:]:
        D.set_field(F[n], -3.0)
    return F


PERIODIC_CASES = [(2, 2, None), (3, 5, None), (10, 10, None), (10, 10, 8), (9, 7, 2), (64, 48, 8), (63, 49, None), (124, 5, 2), (125, 6, 2),
                  (257, 129, None), (256, 130, 64), (300, 77, None), (1021, 33, 64), (130, 260, 64)]


@pytest.mark.parametrize("fallback", [False, True])
@pytest.mark.parametrize("nx,ny,alignment", PERIODIC_CASES)
def test_two_periodic_steps_per_launch_match_the_oracle_model(D, nx, ny, alignment, fallback):
    """dlesm_shallow_step_sw_x2_periodic_f64 against the ORACLE running the periodic model for two steps (orc_sw_step_sw + the
    reference's periodic copies, field_mod.f90:1394-1464): both levels, internal region AND halos, bit for bit; even and odd N
    (the image of a column two cells outside the box lands on either 16-byte phase)"""
    import torch
    L = D._cabi.lib()
    g = _grid_sw(D, nx, ny, alignment)
    F = _periodic_fields(D, g, SEED + 60)
    it = F["p"].internal
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    H = {n: F[n].get_data() for n in NAMES}
    n1 = [H[n].copy() for n in NAMES[6:9]]
    n2 = [H[n].copy() for n in NAMES[9:]]
    O.sw_step_sw(prm, g.nx, it.box(), H["u"], H["v"], H["p"], H["uold"], H["vold"], H["pold"], *n1)
    for f in n1:
        O.apply_periodic_halos(f, g.nx, it.box(), 0, 0)
    O.sw_step_sw(prm, g.nx, it.box(), *n1, H["u"], H["v"], H["p"], *n2)
    for f in n2:
        O.apply_periodic_halos(f, g.nx, it.box(), 0, 0)
    L.dlesm_set_tuning(b"sw_x2_fused", 0 if fallback else 1)
    try:
        D.psy.invoke_shallow_step_sw_x2_periodic(prm, *[F[n] for n in NAMES])
        torch.cuda.synchronize()
    finally:
        L.dlesm_set_tuning(b"sw_x2_fused", 1)
    for name, want in zip(NAMES[6:], n1 + n2):
        got = F[name].get_data()
        a, b = got[:it.ystop + 1, :it.xstop + 1], want[:it.ystop + 1, :it.xstop + 1]      # internal region + the halo ring
        assert np.array_equal(a, b), (name, int(np.count_nonzero(a != b)), np.argwhere(a != b)[:6].tolist())
    for n in NAMES[:6]:
        assert np.array_equal(F[n].get_data(), H[n]), n


@pytest.mark.lab
@pytest.mark.parametrize("nx,ny,alignment", [(10, 10, None), (63, 49, None), (257, 129, None), (1021, 33, 64), (130, 260, 64)])
def test_two_periodic_steps_four_row_tiles(D, nx, ny, alignment):
    """the comparison form of the plain periodic kernel (four-row tiles, registers uncapped: lab build) == the product's form"""
    import torch
    L = D._cabi.lib()
    g = _grid_sw(D, nx, ny, alignment)
    A, B = _periodic_fields(D, g, SEED + 70), _periodic_fields(D, g, SEED + 70)
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    D.psy.invoke_shallow_step_sw_x2_periodic(prm, *[A[n] for n in NAMES])
    L.dlesm_set_tuning(b"sw_x2_sw_form", 1)
    try:
        D.psy.invoke_shallow_step_sw_x2_periodic(prm, *[B[n] for n in NAMES])
        torch.cuda.synchronize()
    finally:
        L.dlesm_set_tuning(b"sw_x2_sw_form", 0)
    for n in NAMES:
        assert torch.equal(A[n].data, B[n].data), n


@pytest.mark.parametrize("fallback", [False, True])
@pytest.mark.parametrize("nx,ny,alignment", [(2, 2, None), (10, 10, None), (9, 7, 2), (64, 48, 8), (63, 49, None), (257, 129, None), (300, 77, None),
                                             (1021, 33, 64), (130, 260, 64)])
def test_two_filtered_periodic_steps_per_launch(D, nx, ny, alignment, fallback):
    _filtered_periodic(D, nx, ny, alignment, fallback, 0)


@pytest.mark.lab
@pytest.mark.parametrize("nx,ny,alignment", [(10, 10, None), (63, 49, None), (257, 129, None), (1021, 33, 64), (130, 260, 64)])
def test_two_filtered_periodic_steps_two_row_tiles(D, nx, ny, alignment):
    """the comparison form of the filtered periodic kernel (two-row tiles, every load issued first: lab build)"""
    _filtered_periodic(D, nx, ny, alignment, False, 1)


def _filtered_periodic(D, nx, ny, alignment, fallback, sw_form):
    """dlesm_shallow_step_sw_smooth_x2_periodic_f64 -- two whole time steps of the GOcean `shallow` loop per launch -- against the
    one-launch filtered periodic step applied twice with the loop's rotation (itself checked against the oracle's loop nests,
    tests/test_gpu_shallow_kernels.py), three double steps in a row: internal region and halos of both levels"""
    import torch
    L = D._cabi.lib()
    g = _grid_sw(D, nx, ny, alignment)
    alpha = 0.001
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    A, B = _periodic_fields(D, g, SEED + 80), _periodic_fields(D, g, SEED + 80)
    it = A["p"].internal
    a_cur, a_old, a_n2, a_o2 = ([A[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6, 9))
    b_cur, b_old, b_new = ([B[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6))
    cut = lambda f: f.data[:it.ystop + 1, :it.xstop + 1]      # noqa: E731
    L.dlesm_set_tuning(b"sw_x2_fused", 0 if fallback else 1)
    if sw_form:
        L.dlesm_set_tuning(b"sw_x2_sw_form", sw_form)
    try:
        for pair in range(3):
            D.psy.invoke_shallow_step_sw_smooth_x2_periodic(prm, alpha, *a_cur, *a_old, *a_n2, *a_o2)
            a_cur, a_old, a_n2, a_o2 = a_n2, a_o2, a_cur, a_old
            for _ in range(2):
                D.psy.invoke_shallow_step_sw_smooth_periodic(prm, alpha, *b_cur, *b_old, *b_new)
                b_cur, b_old, b_new = b_new, b_old, b_cur
            torch.cuda.synchronize()
            for x, y in zip(a_cur + a_old, b_cur + b_old):
                assert torch.equal(cut(x), cut(y)), (pair, int((cut(x) != cut(y)).sum()), (cut(x) != cut(y)).nonzero()[:6].tolist())
    finally:
        L.dlesm_set_tuning(b"sw_x2_fused", 1)
        if sw_form:
            L.dlesm_set_tuning(b"sw_x2_sw_form", 0)


# ---- IEEE special values: subnormals, signed zeros, overflow, infinities, NaN, a zero denominator in z ---------------------------
def _special_fields(D, g):
    """twelve fields with patches of special values in levels n and n-1 (the ring of every level the same, as _fields)"""
    F = _fields(D, g, SEED + 300)
    rng = np.random.default_rng(11)
    H = {n: F[n].get_data() for n in NAMES[:6]}
    for n in NAMES[:6]:
        h = H[n]
        h[6:12, 10:40] = rng.random((6, 30)) * 1e-310          # subnormals
        h[14:17, 20:50] = -0.0
        h[14:17, 50:80] = 0.0
        h[20:23, 30:60] = (1.6e308 if n[0] != "v" else -1.6e308)   # products and sums overflow
        h[30:34, 40:100] = np.ldexp(rng.random((4, 60)), -1060)  # results cross the subnormal boundary
    H["p"][26:29, 20:60] = 0.0                                  # p + p + p + p = 0 under z: x / 0
    H["u"][38, 30] = np.inf
    H["v"][38, 70] = -np.inf
    H["pold"][40, 50] = np.nan
    for n in NAMES[:6]:
        F[n].set_data(H[n])
    it = F["p"].internal
    for k, n in enumerate(NAMES[6:]):                           # every level carries the ring of level n
        D.copy_field(F[NAMES[k % 3]], F[n])
    for k, n in enumerate(NAMES[3:6]):
        keep = F[n].data[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop].clone()
        D.copy_field(F[NAMES[k]], F[n])
        F[n].data[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop] = keep
    return F


def _same_ieee(got, want, what):
    nan_w, nan_g = np.isnan(want), np.isnan(got)
    assert np.array_equal(nan_w, nan_g), (what, "NaN positions differ", int(np.count_nonzero(nan_w != nan_g)))
    a, b = got[~nan_w].view(np.uint64), want[~nan_w].view(np.uint64)
    assert np.array_equal(a, b), (what, int(np.count_nonzero(a != b)))


def test_special_values_follow_ieee_like_the_cpu(D):
    """the fused step, the two-step kernel and the filtered two-step kernel on inputs with subnormals (not flushed), signed
    zeros, overflow, infinities, NaN and a zero denominator: bit-identical to the oracle's loops wherever the result is not a NaN,
    NaN exactly where the oracle has one (the first stage's redundant rim evaluates the same operations on the same operands)"""
    import torch
    g = _grid(D, 140, 48, 8)
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    F = _special_fields(D, g)
    it = F["p"].internal
    box = it.box()
    H = {n: F[n].get_data() for n in NAMES}
    with np.errstate(all="ignore"):
        n1, n2 = _oracle_two_steps(prm, g, box, H)
    D.psy.invoke_shallow_step_x2(prm, *[F[n] for n in NAMES])
    torch.cuda.synchronize()
    for name, want in zip(NAMES[6:], n1 + n2):
        _same_ieee(F[name].get_data(), want, "two steps: " + name)
    assert sum(int(np.count_nonzero(np.isnan(w))) for w in n2) > 0 and sum(int(np.count_nonzero(np.isinf(w))) for w in n1) > 0
    sub = sum(int(np.count_nonzero((np.abs(w) > 0) & (np.abs(w) < 2.3e-308))) for w in n1 + n2)
    assert sub > 50, sub                                        # subnormal results really occur
    # one step, for the record of the single-step kernel
    G = _special_fields(D, g)
    D.psy.invoke_shallow_step(prm, *[G[n] for n in NAMES[:9]])
    torch.cuda.synchronize()
    for name, want in zip(NAMES[6:9], n1):
        _same_ieee(G[name].get_data(), want, "one step: " + name)
    # two filtered steps per launch against two one-launch filtered steps (each checked against the oracle's loop nests elsewhere)
    A, B = _special_fields(D, g), _special_fields(D, g)
    a_cur, a_old, a_n2, a_o2 = ([A[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6, 9))
    for src, dst in zip(a_cur, a_o2):
        D.copy_field(src, dst)
    b_cur, b_old, b_new = ([B[n] for n in NAMES[k:k + 3]] for k in (0, 3, 6))
    D.psy.invoke_shallow_step_smooth_x2(prm, 0.001, *a_cur, *a_old, *a_n2, *a_o2)
    for _ in range(2):
        D.psy.invoke_shallow_step_smooth(prm, 0.001, *b_cur, *b_old, *b_new)
        b_cur, b_new = b_new, b_cur
    torch.cuda.synchronize()
    cut = lambda f: f.get_data()[it.ystart - 1:it.ystop, it.xstart - 1:it.xstop]      # noqa: E731
    for x, y, name in zip(a_n2 + a_o2, b_cur + b_old, NAMES[6:]):
        _same_ieee(cut(x), cut(y), "two filtered steps: " + name)


def test_special_values_in_the_periodic_model(D):
    """the same for the SW-offset doubly periodic kernels (their own expression trees): one-launch step and two steps per launch
    against the oracle's model of the loop (step + the reference's periodic copies), halos included"""
    import torch
    g = _grid_sw(D, 130, 44, 8)
    prm = D.psy.shallow_params(1.0e5, 0.9e5, 40.0)
    F = _periodic_fields(D, g, SEED + 310)
    it = F["p"].internal
    rng = np.random.default_rng(12)
    for n in NAMES[:6]:
        h = F[n].get_data()
        h[6:12, 10:40] = rng.random((6, 30)) * 1e-310
        h[14:17, 20:50] = -0.0
        h[20:23, 30:60] = (1.6e308 if n[0] != "v" else -1.6e308)
        h[30:34, 40:100] = np.ldexp(rng.random((4, 60)), -1060)
        if n == "p":
            h[26:29, 20:60] = 0.0
        if n == "u":
            h[38, 30] = np.inf
        if n == "pold":
            h[40, 50] = np.nan
        F[n].set_data(h)
        D.psy.apply_periodic_halos(F[n])
    torch.cuda.synchronize()
    H = {n: F[n].get_data() for n in NAMES}
    n1 = [H[n].copy() for n in NAMES[6:9]]
    n2 = [H[n].copy() for n in NAMES[9:]]
    with np.errstate(all="ignore"):
        O.sw_step_sw(prm, g.nx, it.box(), H["u"], H["v"], H["p"], H["uold"], H["vold"], H["pold"], *n1)
        for f in n1:
            O.apply_periodic_halos(f, g.nx, it.box(), 0, 0)
        O.sw_step_sw(prm, g.nx, it.box(), *n1, H["u"], H["v"], H["p"], *n2)
        for f in n2:
            O.apply_periodic_halos(f, g.nx, it.box(), 0, 0)
    D.psy.invoke_shallow_step_sw_x2_periodic(prm, *[F[n] for n in NAMES])
    torch.cuda.synchronize()
    for name, want in zip(NAMES[6:], n1 + n2):
        got = F[name].get_data()
        _same_ieee(got[:it.ystop + 1, :it.xstop + 1], want[:it.ystop + 1, :it.xstop + 1], "periodic two steps: " + name)
    assert sum(int(np.count_nonzero(np.isnan(w))) for w in n2) > 0
    G = _periodic_fields(D, g, SEED + 310)
    for n in NAMES[:6]:
        G[n].set_data(H[n])
    D.psy.invoke_shallow_step_sw_periodic(prm, *[G[n] for n in NAMES[:9]])
    torch.cuda.synchronize()
    for name, want in zip(NAMES[6:9], n1):
        got = G[name].get_data()
        _same_ieee(got[:it.ystop + 1, :it.xstop + 1], want[:it.ystop + 1, :it.xstop + 1], "periodic one step: " + name)
