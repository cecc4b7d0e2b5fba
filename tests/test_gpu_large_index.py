"""GPU test (-m gpu): arrays of more than 2^31 ELEMENTS (46400^2 cells, 17 GB per array) -- index arithmetic beyond 32 bits.
The reference's own maximum is whatever fits a default-integer pair (nx, ny); its linear index (jj-1)*nx + (ji-1) exceeds 2^31 from
46341^2 on.  The oracle cannot sweep 2 x 10^9 cells in a test, so the check is structural: row bands cut out of the big array (at the
start, where the linear index crosses 2^30, 2^31 and 2^32 BYTES / ELEMENTS, and at the end) are swept again as SMALL arrays -- the case
the other tests pin to the oracle bit for bit -- and must equal the same rows of the big sweep."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

N = 46400
LD = 46464            # DL_ESM_ALIGNMENT = 64: N + 2 padded
NY = N + 3


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import dl_esm_inf_amd as d
    torch.cuda.set_device(0)
    d.parallel_init(0, 1)
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip(f"needs 200 GB of free device memory, {free / 1e9:.0f} GB there")
    return torch, d, d._cabi.lib()


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _bands():
    """first rows of bands of five rows: the start, the rows where the linear element index crosses 2^28 (2^31 bytes), 2^29,
    2^30, 2^31 elements, and the end"""
    rows = [0, NY - 5]
    for e in (28, 29, 30, 31):
        r = (1 << e) // LD
        rows += [r - 3, r - 1]
    return sorted(r for r in set(rows) if 0 <= r <= NY - 5)


def _big(torch, n, lo):
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + n)
    t = torch.empty((NY, LD), dtype=torch.float64, device="cuda")
    for r0 in range(0, NY, 4096):            # (in slabs: the generator's temporaries stay small)
        t[r0:r0 + 4096].copy_(torch.rand((min(4096, NY - r0), LD), dtype=torch.float64, device="cuda", generator=g) + lo)
    return t


def test_jacobi5_beyond_2_31_elements(T):
    torch, D, L = T
    assert LD * NY > 2 ** 31
    a = _big(torch, 0, 0.0)
    b = torch.full((NY, LD), -7.0, dtype=torch.float64, device="cuda")
    D._cabi.check(L.dlesm_stencil5_f64(_ptr(a), _ptr(b), LD, NY, 2, N + 1, 2, N + 1, None))
    torch.cuda.synchronize()
    for r in _bands():
        sa = a[r:r + 5].contiguous()
        sb = torch.full((5, LD), -7.0, dtype=torch.float64, device="cuda")
        D._cabi.check(L.dlesm_stencil5_f64(_ptr(sa), _ptr(sb), LD, 5, 2, N + 1, 2, 4, None))
        torch.cuda.synchronize()
        lo, hi = max(r + 1, 1), min(r + 4, N + 1)         # rows of the big sweep that are rows 2..4 of the band AND inside its box
        assert torch.equal(b[lo:hi], sb[lo - r:hi - r]), r
    assert bool((b[0] == -7.0).all()) and bool((b[N + 1:] == -7.0).all()) and bool((b[:, 0] == -7.0).all())
    del a
    torch.cuda.empty_cache()


def test_shallow_step_beyond_2_31_elements(T):
    torch, D, L = T
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    ins = [_big(torch, k, 1.0 if k % 3 == 2 else -0.5) for k in range(6)]
    outs = [torch.full((NY, LD), 9.0, dtype=torch.float64, device="cuda") for _ in range(3)]
    D._cabi.check(L.dlesm_shallow_step_f64(C.byref(prm), LD, NY, 2, N + 1, 2, N + 1, *[_ptr(t) for t in ins], *[_ptr(t) for t in outs], None))
    torch.cuda.synchronize()
    for r in _bands():
        si = [t[r:r + 5].contiguous() for t in ins]
        so = [torch.full((5, LD), 9.0, dtype=torch.float64, device="cuda") for _ in range(3)]
        D._cabi.check(L.dlesm_shallow_step_f64(C.byref(prm), LD, 5, 2, N + 1, 2, 4, *[_ptr(t) for t in si], *[_ptr(t) for t in so], None))
        torch.cuda.synchronize()
        lo, hi = max(r + 1, 1), min(r + 4, N + 1)
        for k in range(3):
            assert torch.equal(outs[k][lo:hi], so[k][lo - r:hi - r]), (r, k)
    for k in range(3):
        assert bool((outs[k][0] == 9.0).all()) and bool((outs[k][N + 1:] == 9.0).all())
    del ins, outs
    torch.cuda.empty_cache()


def test_utility_sweeps_beyond_2_31_elements(T):
    """fill, hash_init, copy_patch and the checksum on a box of more than 2^31 cells: the checksum of a constant field is exact
    (value x cells, a power-of-two-friendly value), hash_init equals its small-array self on row bands, the patch copy moves the
    last rows intact"""
    torch, D, L = T
    f = torch.full((NY, LD), -1.0, dtype=torch.float64, device="cuda")
    D._cabi.check(L.dlesm_fill_f64(_ptr(f), LD, NY, 1, LD, 1, NY, C.c_double(0.5), None))
    res = C.c_double(0.0)
    D._cabi.check(L.dlesm_checksum_f64(_ptr(f), LD, NY, 1, LD, 1, NY, C.byref(res), None))
    assert res.value == 0.5 * LD * NY, (res.value, 0.5 * LD * NY)          # exact: every partial sum is a multiple of 0.5 below 2^53
    D._cabi.check(L.dlesm_hash_init_f64(_ptr(f), LD, NY, 2, N + 1, 2, N + 1, C.c_uint64(99), C.c_int64(1), C.c_int64(1), None))
    torch.cuda.synchronize()
    for r in _bands():
        lo, hi = max(r, 1), min(r + 5, N + 1)            # rows of the band inside the box (0-based lo .. hi-1)
        s = torch.full((hi - lo, LD), 0.5, dtype=torch.float64, device="cuda")
        # the same cells as a small array: global row offset = lo (the hash takes global indices)
        D._cabi.check(L.dlesm_hash_init_f64(_ptr(s), LD, hi - lo, 2, N + 1, 1, hi - lo, C.c_uint64(99), C.c_int64(1), C.c_int64(lo + 1), None))
        torch.cuda.synchronize()
        assert torch.equal(f[lo:hi], s), r
    g = torch.zeros((NY, LD), dtype=torch.float64, device="cuda")
    D._cabi.check(L.dlesm_copy_patch_f64(_ptr(f), _ptr(g), LD, NY, 2, N - 2, 2, N - 2, N, 4, None))      # the last four rows of the box
    torch.cuda.synchronize()
    assert torch.equal(g[N - 3:N + 1, 1:N + 1], f[N - 3:N + 1, 1:N + 1]) and float(g[:N - 3].abs().sum()) == 0.0
    del f, g
    torch.cuda.empty_cache()
