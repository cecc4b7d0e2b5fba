"""GPU test (-m gpu): every sweep kernel on boxes that are NOT a field's internal region -- arbitrary
sub-boxes of arbitrary arrays, the way a PSy layer with loop bounds of its own (or the interior of a
distributed step) calls them.  The wave tiles are anchored on 128-byte lines of the row, lanes outside
the box are masked: first columns from 2 to the hundreds, odd and even leading dimensions, one-row and
one-column boxes.  Bit for bit against the oracle; every cell outside the box untouched."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import dl_esm_inf_amd as d
    torch.cuda.set_device(0)
    d.parallel_init(0, 1)
    return torch, d, d._cabi.lib()


def _cases(n, seed):
    """(ld, ny, xs, xe, ys, ye): 1-based inclusive boxes with a one-cell ring inside the array"""
    rng = np.random.default_rng(seed)
    out = [(64, 9, 2, 63, 2, 8), (130, 7, 17, 17, 2, 6), (200, 5, 33, 190, 3, 3), (129, 6, 2, 2, 2, 5),
           (1030, 12, 513, 1029, 2, 11), (300, 40, 129, 140, 5, 30)]
    while len(out) < n:
        ld = int(rng.integers(8, 700))
        ny = int(rng.integers(3, 40))
        xs = int(rng.integers(2, ld - 1))
        xe = int(rng.integers(xs, ld))                      # xe <= ld - 1
        ys = int(rng.integers(2, ny))
        ye = int(rng.integers(ys, ny))                      # ye <= ny - 1
        out.append((ld, ny, xs, xe, ys, ye))
    return out


def _rand(torch, rng, ny, ld, lo=-0.4):
    h = rng.random((ny, ld)) + lo
    return h, torch.from_numpy(h).cuda()


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _inbox(ny, ld, xs, xe, ys, ye):
    m = np.zeros((ny, ld), dtype=bool)
    m[ys - 1:ye, xs - 1:xe] = True
    return m


@pytest.mark.parametrize("case", _cases(40, 1), ids=lambda c: "x".join(map(str, c)))
def test_two_and_three_stream_sweeps_on_subboxes(T, case):
    torch, D, L = T
    ld, ny, xs, xe, ys, ye = case
    rng = np.random.default_rng(ld * 1000 + ny)
    hin, din = _rand(torch, rng, ny, ld)
    coef = rng.random(9) - 0.45
    mask = (rng.random((ny, ld)) > 0.25).astype(np.int32)
    dmask = torch.from_numpy(mask).cuda()
    for name in ("jacobi5", "stencil9", "masked"):
        hout = np.full((ny, ld), -7.0)
        dout = torch.full((ny, ld), -7.0, dtype=torch.float64, device="cuda")
        if name == "jacobi5":
            D._cabi.check(L.dlesm_stencil5_f64(_ptr(din), _ptr(dout), ld, ny, xs, xe, ys, ye, None))
            O.jacobi5(hin, hout, ld, xs, xe, ys, ye)
        elif name == "stencil9":
            D._cabi.check(L.dlesm_stencil9_f64(_ptr(din), _ptr(dout), coef.ctypes.data_as(C.POINTER(C.c_double)), ld, ny,
                                               xs, xe, ys, ye, None))
            O.stencil9(hin, hout, coef, ld, xs, xe, ys, ye)
        else:
            D._cabi.check(L.dlesm_stencil5_masked_f64(_ptr(din), _ptr(dout), _ptr(dmask), ld, ny, xs, xe, ys, ye, None))
            O.jacobi5_masked(hin, hout, mask, ld, xs, xe, ys, ye)
        torch.cuda.synchronize()
        assert np.array_equal(dout.cpu().numpy(), hout), name


@pytest.mark.parametrize("case", _cases(30, 2), ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("sw_offset", [False, True])
def test_shallow_and_continuity_on_subboxes(T, case, sw_offset):
    torch, D, L = T
    ld, ny, xs, xe, ys, ye = case
    rng = np.random.default_rng(ld * 77 + ny)
    H, Dv = zip(*[_rand(torch, rng, ny, ld, lo=(1.0 if k == 2 or k == 5 else -0.5)) for k in range(6)])
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    hnew = [np.full((ny, ld), 9.0) for _ in range(3)]
    dnew = [torch.full((ny, ld), 9.0, dtype=torch.float64, device="cuda") for _ in range(3)]
    fn = L.dlesm_shallow_step_sw_f64 if sw_offset else L.dlesm_shallow_step_f64
    D._cabi.check(fn(C.byref(prm), ld, ny, xs, xe, ys, ye, *[_ptr(t) for t in Dv], *[_ptr(t) for t in dnew], None))
    (O.sw_step_sw if sw_offset else O.sw_step)(prm, ld, (xs, xe, ys, ye), *H, *hnew)
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(dnew[k].cpu().numpy(), hnew[k]), ("unew", "vnew", "pnew")[k]
    if sw_offset:
        return
    # continuity: eight inputs, of which the six above + two more
    (h6, d6), (h7, d7) = _rand(torch, rng, ny, ld), _rand(torch, rng, ny, ld, lo=0.5)
    hs = np.full((ny, ld), -7.0)
    ds = torch.full((ny, ld), -7.0, dtype=torch.float64, device="cuda")
    D._cabi.check(L.dlesm_continuity_f64(0.37, ld, ny, xs, xe, ys, ye, *[_ptr(t) for t in Dv], _ptr(d6), _ptr(d7), _ptr(ds),
                                         None))
    O.continuity(0.37, ld, (xs, xe, ys, ye), *H, h6, h7, hs)
    torch.cuda.synchronize()
    assert np.array_equal(ds.cpu().numpy(), hs)


@pytest.mark.parametrize("case", _cases(36, 3), ids=lambda c: "x".join(map(str, c)))
def test_two_steps_per_launch_on_subboxes(T, case):
    """dlesm_shallow_step_x2_f64 on an arbitrary box of arbitrary arrays == the oracle's step twice on that box (the second step
    reads level n+1 one cell outside the box from where the level-n+1 arrays hold it); cells outside the box untouched"""
    torch, D, L = T
    ld, ny, xs, xe, ys, ye = case
    if xs < 2 or ys < 2 or xe > ld - 1 or ye > ny - 1:
        pytest.skip("no room for the stencil ring")
    rng = np.random.default_rng(ld * 131 + ny)
    H, Dv = zip(*[_rand(torch, rng, ny, ld, lo=(1.0 if k % 3 == 2 else -0.5)) for k in range(12)])
    H = [h.copy() for h in H]
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 40.0)
    D._cabi.check(L.dlesm_shallow_step_x2_f64(C.byref(prm), ld, ny, xs, xe, ys, ye, *[_ptr(t) for t in Dv], None))
    box = (xs, xe, ys, ye)
    O.sw_step(prm, ld, box, *H[:6], *H[6:9])
    O.sw_step(prm, ld, box, *H[6:9], *H[:3], *H[9:])
    torch.cuda.synchronize()
    for k in range(12):
        assert np.array_equal(Dv[k].cpu().numpy(), H[k]), k


@pytest.mark.parametrize("case", _cases(30, 4), ids=lambda c: "x".join(map(str, c)))
def test_two_filtered_steps_per_launch_on_subboxes(T, case):
    """dlesm_shallow_step_smooth_x2_f64 on an arbitrary box: the wave-tile kernel against the entry's own definition (the path
    arrays that miss the tile conditions take: two one-launch filtered steps through scratch copies), every array, every cell"""
    torch, D, L = T
    ld, ny, xs, xe, ys, ye = case
    rng = np.random.default_rng(ld * 17 + ny)
    H, A = zip(*[_rand(torch, rng, ny, ld, lo=(1.0 if k % 3 == 2 else -0.5)) for k in range(12)])
    B = [t.clone() for t in A]
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 40.0)
    D._cabi.check(L.dlesm_shallow_step_smooth_x2_f64(C.byref(prm), 0.001, ld, ny, xs, xe, ys, ye, *[_ptr(t) for t in A], None))
    L.dlesm_set_tuning(b"sw_x2_fused", 0)
    try:
        D._cabi.check(L.dlesm_shallow_step_smooth_x2_f64(C.byref(prm), 0.001, ld, ny, xs, xe, ys, ye, *[_ptr(t) for t in B], None))
    finally:
        L.dlesm_set_tuning(b"sw_x2_fused", 1)
    torch.cuda.synchronize()
    for k in range(12):
        if k < 6:
            assert np.array_equal(A[k].cpu().numpy(), H[k]), ("input modified", k)
        if k < 6:
            continue
        # the box is what both forms define (outside it the definition leaves a copy of the old level in uold2.., the kernel
        # what was there: a time loop keeps every array's ring)
        a, b = A[k][ys - 1:ye, xs - 1:xe], B[k][ys - 1:ye, xs - 1:xe]
        assert torch.equal(a, b), (k, int((a != b).sum()))
        if k < 9:      # level n+2 is written on the box only, by both
            assert np.array_equal(np.where(_inbox(ny, ld, xs, xe, ys, ye), 0.0, A[k].cpu().numpy()),
                                  np.where(_inbox(ny, ld, xs, xe, ys, ye), 0.0, H[k])), ("cells outside the box written", k)
