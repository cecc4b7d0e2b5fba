"""GPU tests (-m gpu): 2, 3 and 4 fused Jacobi steps (dlesm_stencil5_x2_f64 / _multi_f64) against
the same number of applications of the oracle's single step through ping-pong buffers, bit for
bit; and, at full size, against single GPU steps."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 20261004


@pytest.fixture(scope="module")
def D():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import dl_esm_inf_amd as d
    torch.cuda.set_device(0)
    d.parallel_init(0, 1)
    return d


def _grid(D, nx, ny, alignment):
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    return g


def _oracle_multi(hin, ld, nsteps, box, ebox, grow=(0, 0, 0, 0)):
    """t_s = J(t_{s-1}) on E_s (t_{s-1} elsewhere), E_s = ebox grown by (nsteps-1-s)*grow;
    out = J(t_{nsteps-1}) on box (left at -7 elsewhere)"""
    t = hin.copy()
    for s in range(1, nsteps):
        k = nsteps - 1 - s
        e = (ebox[0] - grow[0] * k, ebox[1] + grow[1] * k, ebox[2] - grow[2] * k, ebox[3] + grow[3] * k)
        nxt = t.copy()
        if ebox[0] <= ebox[1] and ebox[2] <= ebox[3]:
            O.jacobi5(t, nxt, ld, *e)
        t = nxt
    want = np.full_like(hin, -7.0)
    O.jacobi5(t, want, ld, *box)
    return want


CASES = [(1, 1, None), (2, 3, 2), (4, 10, None), (10, 4, 8), (5, 5, 8), (64, 64, 64), (61, 67, None), (61, 67, 2),
         (119, 9, 2), (120, 9, 2), (121, 9, 2), (123, 9, 2), (124, 9, 2), (125, 9, 2), (247, 31, 64), (248, 31, 64),
         (249, 31, 64), (255, 130, 64), (256, 256, None), (511, 70, 4), (1021, 33, 64), (1500, 200, 64),
         (1500, 200, None), (4096, 300, 64), (1930, 37, 2), (2500, 40, 64), (3001, 21, None)]
STEPS = [2, 3, 4, 5, 6, 7, 8]
TUNES = [dict(), dict(j5xt_rows=12), dict(j5xt_rows=16, j5xt_dpp=0), dict(j5xt_rows=2, j5xt_dpp=0), dict(j5xt_rows=8, j5_tpb=8), dict(j5xt_rows=4, j5_tpb=2, j5xt_dpp=0),
         dict(j5xt_march=1), dict(j5xt_march=1, j5xt_march_ring=6, j5xt_dpp=0),
         dict(j5xt_march=1, j5xt_march_ring=12, j5xt_march_slots=40), dict(j5xt_march=1, j5xt_march_slots=100000),
         dict(j5_variant=4)]
DEFAULTS = dict(j5xt_rows=0, j5xt_dpp=1, j5_tpb=0, j5_variant=0, j5xt_march=0, j5xt_march_ring=9,
                j5xt_march_slots=3072)
# wide and tall enough for the marching kernel to take the interior (8 wave tiles, 4T rows)
CASES += [(1100, 90, 64), (1100, 90, None), (2047, 150, 2), (5000, 64, 64)]


def _tune(D, kw):
    for k, v in kw.items():
        D._cabi.lib().dlesm_set_tuning(k.encode(), v)


# the forms the product library holds run the whole matrix; the comparison forms of the lab build (tile heights, shuffles instead
# of DPP, the pipeline form) a representative part of it -- they are measurements' counterparts, not what ships
LAB_CASES = [(1, 1, None), (10, 4, 8), (61, 67, None), (124, 9, 2), (249, 31, 64), (1021, 33, 64), (1100, 90, None), (2047, 150, 2), (5000, 64, 64)]
LAB_STEPS = [2, 5, 8]


def _matrix():
    from conftest import needs_lab
    out = []
    for tune in TUNES:
        lab = needs_lab({**DEFAULTS, **tune})
        for case in (LAB_CASES if lab else CASES):
            for nsteps in (LAB_STEPS if lab else STEPS):
                tid = "-".join(f"{k}{v}" for k, v in tune.items()) or "default"
                out.append(pytest.param(*case, nsteps, tune, id=f"{tid}-{nsteps}-{case[0]}-{case[1]}-{case[2]}"))
    return out


@pytest.mark.parametrize("nx,ny,alignment,nsteps,tune", _matrix())
def test_fused_steps_bit_exact(D, nx, ny, alignment, nsteps, tune):
    _tune(D, {**DEFAULTS, **tune})
    try:
        g = _grid(D, nx, ny, alignment)
        a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
        D.psy.hash_init(a, SEED)
        D.set_field(b, -7.0)
        if nsteps == 2 and not tune:
            D.psy.invoke_jacobi5_x2(b, a)
        else:
            D.psy.invoke_jacobi5_multi(b, a, nsteps)
        hin, got = a.get_data(), b.get_data()
        box = b.internal.box()
        want = _oracle_multi(hin, g.nx, nsteps, box, box)
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    finally:
        _tune(D, DEFAULTS)


@pytest.mark.parametrize("march", [0, 1], ids=["tile", "march"])      # (march: the pipeline form, libdlesm_hip_lab.so)
@pytest.mark.parametrize("nsteps", STEPS)
def test_fused_sub_boxes_and_grown_stage_boxes(D, nsteps, march):
    """the forms the distributed step uses: thin output boxes with the tile's stage boxes, stage
    boxes grown towards some sides (deep halos), empty boxes; refusals"""
    L = D._cabi.lib()
    import torch
    g = _grid(D, 1216, 105, 2)          # 1220 x 108 array; cells 9..1208 x 9..98 play the tile interior
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    D.psy.hash_init(a, SEED, box=D._cabi.Region(0, 0, 1, g.nx, 1, g.ny))
    hin = a.get_data()
    assert g.nx >= 1217 and g.ny >= 107
    x0, x1, y0, y1 = full = (9, 1208, 9, 98)
    allg = (x0 - 1, x1 + 1, y0 - 1, y1 + 1)   # last stage box of a tile with neighbours on every side
    T = nsteps
    cases = [
        (full, full, (0, 0, 0, 0)), (full, allg, (1, 1, 1, 1)), (full, (x0 - 1, x1, y0, y1 + 1), (1, 0, 0, 1)),
        (full, (x0, x1 + 1, y0 - 1, y1), (0, 1, 1, 0)),
        ((x0, x0 + T - 1, y0, y1), allg, (1, 1, 1, 1)), ((x1 - T + 1, x1, y0, y1), allg, (1, 1, 1, 1)),
        ((x0, x1, y0, y0 + T - 1), allg, (1, 1, 1, 1)), ((x0, x1, y1 - T + 1, y1), (x0 - 1, x1, y0 - 1, y1 + 1), (1, 0, 1, 1)),
        ((x0 + T, x1 - T, y0 + T, y1 - T), allg, (1, 1, 1, 1)), ((17, 17, 40, 40), full, (0, 0, 0, 0)),
        ((40, 39, y0, y1), full, (0, 0, 0, 0)), (full, (10, 9, y0, y1), (0, 0, 0, 0)), ((x0, x1, 50, 49), full, (1, 1, 1, 1)),
    ]
    for box, ebox, grow in cases:
        want = _oracle_multi(hin, g.nx, nsteps, box, ebox, grow)
        if True:                        # march = 0: tile kernel everywhere / 1: marching kernel in the interior (lab build)
            _tune(D, dict(j5xt_march=march, j5xt_march_slots=64) if march else {})
            D.set_field(b, -7.0)
            D._cabi.check(L.dlesm_stencil5_multi_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, nsteps, *box, *ebox,
                                                     *grow, None))
            torch.cuda.synchronize()
            got = b.get_data()
            _tune(D, DEFAULTS)
            assert np.array_equal(got, want), (box, ebox, grow, march, np.argwhere(got != want)[:5])
    # boxes whose stencil ring leaves the array are refused, so are aliased arrays and bad step counts
    k = nsteps - 2
    bad = [((1, 10, y0, 10), full, (0, 0, 0, 0)), (full, (k + 1, x1 + 1, y0 - 1, y1 + 1), (1, 1, 1, 1)),
           (full, (x0 - 1, g.nx - k, y0 - 1, y1 + 1), (1, 1, 1, 1)), (full, allg, (2, 0, 0, 0))]
    for box, ebox, grow in bad:
        rc = L.dlesm_stencil5_multi_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, nsteps, *box, *ebox, *grow, None)
        assert rc == D._cabi.EINVAL, (box, ebox, grow)
    assert L.dlesm_stencil5_multi_f64(a.device_ptr, a.device_ptr, g.nx, g.ny, nsteps, *full, *full, 0, 0, 0, 0,
                                      None) == D._cabi.EINVAL
    for n in (0, 1, 9):
        assert L.dlesm_stencil5_multi_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, n, *full, *full, 0, 0, 0, 0,
                                          None) == D._cabi.EINVAL


def test_x2_entry_with_explicit_intermediate_box(D):
    L = D._cabi.lib()
    g = _grid(D, 302, 91, 2)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    D.psy.hash_init(a, SEED, box=D._cabi.Region(0, 0, 1, g.nx, 1, g.ny))
    hin = a.get_data()
    full = (3, 302, 3, 91)
    for box, ebox in [(full, full), (full, (2, 303, 2, 92)), ((3, 4, 3, 91), (2, 303, 2, 92)),
                      ((5, 300, 5, 89), (2, 303, 2, 92)), (full, (10, 9, 3, 91))]:
        D.set_field(b, -7.0)
        D._cabi.check(L.dlesm_stencil5_x2_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, *box, *ebox, None))
        want = _oracle_multi(hin, g.nx, 2, box, ebox)
        assert np.array_equal(b.get_data(), want), (box, ebox)
    assert L.dlesm_stencil5_x2_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, *full, 1, 303, 2, 92, None) == D._cabi.EINVAL


@pytest.mark.parametrize("march", [0, 1], ids=["tile", "march"])      # (march: the pipeline form, libdlesm_hip_lab.so)
@pytest.mark.parametrize("n,alignment", [(4096, 64), (8192, None), (16384, 64)])
def test_fused_equals_single_steps_full_size(D, n, alignment, march):
    import torch
    g = _grid(D, n, n, alignment)
    a, p, q, c = (D.r2d_field(g, D.GO_T_POINTS) for _ in range(4))
    D.psy.hash_init(a, SEED)
    for f in (p, q, c):
        D.copy_field(a, f)              # same fixed ring everywhere
    src, dst = a, p
    for nsteps in (1, 2, 3, 4, 5, 6, 7, 8):
        D.psy.invoke_jacobi5(dst, src)  # dst = J^nsteps(a)
        if True:
            if nsteps > 1:
                _tune(D, dict(j5xt_march=march) if march else {})
                D.psy.invoke_jacobi5_multi(c, a, nsteps)
                torch.cuda.synchronize()
                _tune(D, DEFAULTS)
                assert torch.equal(dst.data, c.data), (nsteps, march)
        src, dst = dst, (q if dst is p else p)
