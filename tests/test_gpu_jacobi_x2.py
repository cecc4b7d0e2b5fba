"""GPU tests (-m gpu): two fused Jacobi steps (dlesm_stencil5_x2_f64) against two applications of
the oracle's single step through a ping-pong buffer, bit for bit; and, at full size, against two
single GPU steps."""
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


def _oracle_x2(hin, ld, box, ebox):
    """t = J(in) on ebox (in elsewhere); out = J(t) on box (left at -7 elsewhere)"""
    t = hin.copy()
    if ebox[0] <= ebox[1] and ebox[2] <= ebox[3]:
        O.jacobi5(hin, t, ld, *ebox)
    want = np.full_like(hin, -7.0)
    O.jacobi5(t, want, ld, *box)
    return want


CASES = [(1, 1, None), (2, 3, 2), (4, 10, None), (10, 4, 8), (5, 5, 8), (64, 64, 64), (61, 67, None), (61, 67, 2),
         (123, 9, 2), (124, 9, 2), (125, 9, 2), (247, 31, 64), (248, 31, 64), (249, 31, 64), (255, 130, 64),
         (256, 256, None), (511, 70, 4), (1021, 33, 64), (1500, 200, 64), (1500, 200, None), (4096, 300, 64)]
TUNES = [dict(), dict(j5x2_tile_rows=2), dict(j5x2_tile_rows=3, j5_tpb=2), dict(j5x2_tile_rows=6, j5_tpb=8),
         dict(j5x2_tile_rows=8), dict(j5_variant=4)]
DEFAULTS = dict(j5x2_tile_rows=4, j5_tpb=0, j5_variant=0)


@pytest.mark.parametrize("nx,ny,alignment", CASES)
@pytest.mark.parametrize("tune", TUNES, ids=lambda t: "-".join(f"{k}{v}" for k, v in t.items()) or "default")
def test_fused_two_steps_bit_exact(D, nx, ny, alignment, tune):
    L = D._cabi.lib()
    for k, v in {**DEFAULTS, **tune}.items():
        L.dlesm_set_tuning(k.encode(), v)
    try:
        g = _grid(D, nx, ny, alignment)
        a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
        D.psy.hash_init(a, SEED)
        D.set_field(b, -7.0)
        D.psy.invoke_jacobi5_x2(b, a)
        hin, got = a.get_data(), b.get_data()
        box = b.internal.box()
        want = _oracle_x2(hin, g.nx, box, box)
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    finally:
        for k, v in DEFAULTS.items():
            L.dlesm_set_tuning(k.encode(), v)


def test_fused_sub_boxes_and_grown_intermediate_box(D):
    """the forms the distributed step uses: thin output boxes with the full intermediate box,
    an intermediate box grown by one cell on some sides (depth-2 halos), empty boxes"""
    L = D._cabi.lib()
    import torch
    g = _grid(D, 302, 91, 2)            # 306 x 94 array; cells 3..302 x 3..91 play the tile interior
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    D.psy.hash_init(a, SEED, box=D._cabi.Region(0, 0, 1, g.nx, 1, g.ny))
    hin = a.get_data()
    assert g.nx >= 304 and g.ny >= 93
    full = (3, 302, 3, 91)              # interior of a tile with depth-2 halos
    cases = [
        (full, full), (full, (2, 303, 2, 92)), (full, (2, 302, 3, 92)), (full, (3, 303, 2, 91)),
        ((3, 4, 3, 91), (2, 303, 2, 92)), ((301, 302, 3, 91), (2, 303, 2, 92)),
        ((3, 302, 3, 4), (2, 303, 2, 92)), ((3, 302, 90, 91), (2, 302, 2, 92)),
        ((5, 300, 5, 89), (2, 303, 2, 92)), ((17, 17, 40, 40), full), ((40, 39, 3, 91), full),
        (full, (10, 9, 3, 91)), ((3, 302, 50, 49), full),
    ]
    for box, ebox in cases:
        D.set_field(b, -7.0)
        D._cabi.check(L.dlesm_stencil5_x2_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, *box, *ebox, None))
        torch.cuda.synchronize()
        want = _oracle_x2(hin, g.nx, box, ebox)
        got = b.get_data()
        assert np.array_equal(got, want), (box, ebox, np.argwhere(got != want)[:5])
    # boxes whose stencil ring leaves the array are refused
    for box, ebox in [((1, 10, 3, 10), full), (full, (1, 303, 2, 92)), (full, (2, g.nx, 2, 92))]:
        rc = L.dlesm_stencil5_x2_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, *box, *ebox, None)
        assert rc == D._cabi.EINVAL
    assert L.dlesm_stencil5_x2_f64(a.device_ptr, a.device_ptr, g.nx, g.ny, *full, *full, None) == D._cabi.EINVAL


@pytest.mark.parametrize("n,alignment", [(4096, 64), (8192, None), (16384, 64)])
def test_fused_equals_two_single_steps_full_size(D, n, alignment):
    import torch
    g = _grid(D, n, n, alignment)
    a, b, c = (D.r2d_field(g, D.GO_T_POINTS) for _ in range(3))
    D.psy.hash_init(a, SEED)
    D.copy_field(a, b)
    D.copy_field(a, c)
    D.psy.invoke_jacobi5(b, a)          # b = J(a), ring of b = ring of a
    two = D.r2d_field(g, D.GO_T_POINTS)
    D.copy_field(a, two)
    D.psy.invoke_jacobi5(two, b)        # two = J(J(a))
    D.psy.invoke_jacobi5_x2(c, a)
    torch.cuda.synchronize()
    assert torch.equal(two.data, c.data)
