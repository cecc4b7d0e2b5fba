"""Pin the oracle (oracle/dlesm_oracle.c) before anything trusts it.

Bit-exact against tests/golden/ref_*.json -- outputs of the REAL reference run in
the build container by oracle/make_golden.py -- and against the reference's own
known-answer tests (test_halos / test_gsum / test_reduction), restated in
tests/ref_cases.py.  CPU only.
"""
import numpy as np
import pytest

import oracle_lib as O
import ref_cases as R
from conftest import load_golden


# --------------------------------------------------------------------------- decomposition
@pytest.mark.parametrize("case", load_golden("ref_decomp")["cases"],
                         ids=lambda c: f"{c['domainx']}x{c['domainy']}n{c['ndomains']}")
def test_decompose_matches_reference(case):
    d, subs = O.decompose(case["domainx"], case["domainy"], case["ndomains"])
    assert [d.global_nx, d.global_ny, d.nx, d.ny, d.ndomains, d.max_width, d.max_height] == \
           [case["global_nx"], case["global_ny"], case["nx"], case["ny"], case["ndom_out"],
            case["max_width"], case["max_height"]]
    got = [s.glob.as6() + s.internal.as6() for s in subs]
    assert got == case["subdomains"]


# --------------------------------------------------------------------------- extents + bounds
def test_extents_and_bounds_match_reference():
    cases = load_golden("ref_bounds")["cases"]
    assert len(cases) == 700
    n_abort = 0
    for c in cases:
        d, subs = O.decompose(c["nx"], c["ny"], 1)
        sub = subs[0]
        gnx, gny = O.grid_extents(sub.glob.nx, sub.glob.ny, c["alignment"])
        assert [gnx, gny, d.global_nx, d.global_ny] == c["grid"], c
        rc, internal, whole = O.field_bounds(c["ptype"], c["offset"], c["bcx"], c["bcy"],
                                             sub.internal, gnx, gny)
        assert bool(rc) == c["abort"], c
        if c["abort"]:
            n_abort += 1
            continue
        assert internal.as6() == c["internal"], c
        assert whole.as6() == c["whole"], c
        assert c["shape"] == [gnx, gny]           # every field is allocated grid%nx x grid%ny
    assert n_abort == 180


def test_alignment_formula_large():
    # A=64 -> N=4096: 4160, 8192: 8256, 16384: 16448 ; A=1: N+3
    for n, want in [(4096, 4160), (8192, 8256), (16384, 16448)]:
        assert O.grid_extents(n + 2, n + 2, 64) == (want, n + 3)
        assert O.grid_extents(n + 2, n + 2, None) == (n + 3, n + 3)


# --------------------------------------------------------------------------- config-1 plumbing
def test_model_and_gather_match_reference():
    g = load_golden("ref_model")
    assert g["example_4x10"] == {"U": 40.0, "V": 40.0, "T": 40.0, "F": 40.0}
    for m in g["model"]:
        d, subs = O.decompose(m["nx"], m["ny"], 1)
        gnx, gny = O.grid_extents(subs[0].glob.nx, subs[0].glob.ny)
        rc, internal, _ = O.field_bounds(R.GO_T, R.OFFSET_NE, R.BC_EXTERNAL, R.BC_EXTERNAL,
                                         subs[0].internal, gnx, gny)
        assert rc == 0 and [gnx, gny] == m["grid"][:2]
        assert internal.as6()[:4] == m["internal"]
        f = np.full((gny, gnx), m["fill"])
        cs = O.lib().orc_checksum(f, gnx, *internal.as6()[:4])
        assert cs == m["checksum"]
        xt, yt = R.t_coords(1, 1, 2, 2, gnx, gny)
        assert [xt[0], xt[1], xt[-1]] == m["xt"] and [yt[0], yt[1], yt[-1]] == m["yt"]
    for m in g["gather"]:
        d, subs = O.decompose(m["nx"], m["ny"], 1)
        gnx, gny = O.grid_extents(subs[0].glob.nx, subs[0].glob.ny)
        glob = R.unique_global(m["nx"], m["ny"])
        loc = np.zeros((gny, gnx))
        O.lib().orc_scatter(glob, m["nx"], subs[0], loc, gnx)
        assert [loc[0, 0], loc[1, 1], loc[m["ny"], m["nx"]], loc[m["ny"] + 1, m["nx"] + 1]] == m["corner"]
        assert O.lib().orc_checksum(loc, gnx, 2, m["nx"] + 1, 2, m["ny"] + 1) == m["checksum"]
        back = O.gather_all([loc], [gnx], d, subs)
        assert back.shape == tuple(m["gather_shape"][::-1])
        assert np.array_equal(back, glob) and m["gather_mismatch"] == 0


# --------------------------------------------------------------------------- message tables
def _setup(nx, ny, nranks):
    d, subs = O.decompose(nx, ny, nranks)
    comms = [O.map_comms(d, subs, nranks, r + 1) for r in range(nranks)]
    ext = [O.grid_extents(s.glob.nx, s.glob.ny) for s in subs]
    return d, subs, comms, ext


def test_map_comms_matches_survey_probe():
    g = load_golden("survey_probe_mapcomms")
    d, subs, comms, ext = _setup(g["domainx"], g["domainy"], g["nranks"])
    assert (d.nx, d.ny) == (g["ntilex"], g["ntiley"]) and ext[0] == (g["grid_nx"], g["grid_ny"])
    c1 = comms[0]
    assert c1.nsend == g["rank1"]["nsend"] and c1.nrecv == g["rank1"]["nrecv"]
    for got, want in zip(c1.sends(), g["rank1"]["sends"]):
        assert {k: got[k] for k in want} == want
    for got, want in zip(c1.recvs(), g["rank1"]["recvs"]):
        assert {k: got[k] for k in want} == want
    c4 = comms[3]
    assert c4.nsend == 5 and c4.nrecv == 5
    assert [[s["nx"], s["ny"]] for s in c4.sends()] == g["rank4"]["send_shapes"]


@pytest.mark.parametrize("nx,ny,nranks", R.HALO_CASES + [(16, 32, 8), (13, 13, 9), (64, 64, 16)])
def test_map_comms_is_self_consistent(nx, ny, nranks):
    """every send has exactly one matching receive of the same size (tag = dir) and lands on a
    halo cell of the receiver; sources are internal cells of the sender"""
    d, subs, comms, ext = _setup(nx, ny, nranks)
    for r, c in enumerate(comms):
        for s in c.sends():
            peer = comms[s["dest"]]
            m = [q for q in peer.recvs() if q["src"] == r and q["dir"] == s["dir"]]
            assert len(m) == 1
            q = m[0]
            assert (q["nx"], q["ny"]) == (s["nx"], s["ny"])
            assert (q["ides"], q["jdes"]) == (s["ides"], s["jdes"])
            it = subs[r].internal
            assert it.xstart <= s["isrc"] and s["isrc"] + s["nx"] - 1 <= it.xstop
            assert it.ystart <= s["jsrc"] and s["jsrc"] + s["ny"] - 1 <= it.ystop
            ot = subs[s["dest"]].internal
            inside_x = ot.xstart <= q["ides"] <= ot.xstop
            inside_y = ot.ystart <= q["jdes"] <= ot.ystop
            assert not (inside_x and inside_y)


# --------------------------------------------------------------------------- reference's dist_mem tests
@pytest.mark.parametrize("nx,ny,nranks", R.HALO_CASES + [(16, 32, 8), (13, 13, 9)])
def test_halo_exchange_known_answer(nx, ny, nranks):
    """tests/dist_mem/test_halos.f90 on the oracle, for U, V, T and F fields"""
    d, subs, comms, ext = _setup(nx, ny, nranks)
    for ptype in (R.GO_T, R.GO_U, R.GO_V, R.GO_F):
        fields, before = [], []
        for r, s in enumerate(subs):
            it = s.internal.as6()[:4]              # NE + external: every type = subdomain internal
            f = R.init_field_hill(ptype, ext[r][0], ext[r][1], it, s.glob.xstart, s.glob.ystart)
            fields.append(f)
            before.append(f.copy())
        assert O.exchange_all(fields, [e[0] for e in ext], comms) == 0
        for r, s in enumerate(subs):
            it = s.internal.as6()[:4]
            bad = R.check_hill_halos(fields[r], ptype, it, s.glob.as6()[:4], nx, ny, corners=True)
            assert not bad, (r, ptype, bad[:3])
            # internal cells untouched
            xs, xe, ys, ye = it
            assert np.array_equal(fields[r][ys - 1:ye, xs - 1:xe], before[r][ys - 1:ye, xs - 1:xe])
            # cells outside the depth-1 ring untouched
            assert np.array_equal(fields[r][ye + 1:, :], before[r][ye + 1:, :])
            assert np.array_equal(fields[r][:, xe + 1:], before[r][:, xe + 1:])


@pytest.mark.parametrize("nx,ny,nranks", R.GSUM_CASES)
def test_gsum_known_answer(nx, ny, nranks):
    """tests/dist_mem/test_gsum.f90: global checksum == jpiglo*jpjglo"""
    d, subs, comms, ext = _setup(nx, ny, nranks)
    total = 0.0
    for r, s in enumerate(subs):
        it = s.internal.as6()[:4]
        f = R.gsum_field(ext[r][0], ext[r][1], it)
        total += O.lib().orc_checksum(f, ext[r][0], *it)
    assert total == float(nx * ny)


@pytest.mark.parametrize("nx,ny,nranks", R.REDUCTION_CASES)
def test_scatter_gather_known_answer(nx, ny, nranks):
    """tests/dist_mem/test_reduction.f90: scatter is index-exact, gather(value+1) round-trips"""
    d, subs, comms, ext = _setup(nx, ny, nranks)
    glob = R.unique_global(nx, ny)
    fields = []
    for r, s in enumerate(subs):
        f = np.zeros((ext[r][1], ext[r][0]))
        O.lib().orc_scatter(glob, nx, s, f, ext[r][0])
        xs, xe, ys, ye = s.internal.as6()[:4]
        want = glob[s.glob.ystart - 1:s.glob.ystop, s.glob.xstart - 1:s.glob.xstop]
        assert np.array_equal(f[ys - 1:ye, xs - 1:xe], want)
        f[ys - 1:ye, xs - 1:xe] += 1.0
        fields.append(f)
    back = O.gather_all(fields, [e[0] for e in ext], d, subs)
    assert np.array_equal(back, glob + 1.0)


# --------------------------------------------------------------------------- stencil self-checks
def test_hash_init_twin():
    f = O.hash_field(20261004, 9, 12, 5, 7, 2, 8, 2, 6)
    for (j, i) in [(2, 2), (6, 8), (3, 5)]:
        assert f[j - 1, i - 1] == O.lib().orc_hash_u01(20261004, 5 + i - 1, 7 + j - 1)
    assert f[0, 0] == 0.0 and 0.0 <= f.min() and f.max() < 1.0


def test_jacobi5_against_numpy():
    """PARITY UNPINNED by the reference (it has no stencil); independent numpy evaluation"""
    n, ld = 37, 44
    rng = np.random.default_rng(1)
    a = rng.random((n + 3, ld))
    out = np.full_like(a, -7.0)
    O.jacobi5(a, out, ld, 2, n + 1, 2, n + 1)
    want = 0.25 * ((a[1:n + 1, 0:n] + a[1:n + 1, 2:n + 2]) + (a[0:n, 1:n + 1] + a[2:n + 2, 1:n + 1]))
    assert np.array_equal(out[1:n + 1, 1:n + 1], want)
    out[1:n + 1, 1:n + 1] = -7.0
    assert np.all(out == -7.0)                    # nothing outside the box is written
    out2 = np.full_like(a, -7.0)
    O.jacobi5(a, out2, ld, 2, n + 1, 2, n + 1, threads=3)
    assert np.array_equal(out2[1:n + 1, 1:n + 1], want)


def test_fortran_psy_loops_equal_the_c_oracle():
    """the same step written the way a GOcean application runs it on the CPU -- Fortran pointwise
    kernel called from the PSy loop nest, with and without OpenMP over jj -- is bit-identical"""
    rng = np.random.default_rng(3)
    for ny, ld, box in [(40, 44, (2, 38, 2, 37)), (9, 200, (2, 199, 2, 8)), (130, 17, (3, 9, 5, 120)),
                        (5, 5, (3, 3, 3, 3))]:
        a = rng.random((ny, ld))
        want = np.full_like(a, -7.0)
        O.jacobi5(a, want, ld, *box)
        for threads in (1, 4):
            got = np.full_like(a, -7.0)
            O.jacobi5_fortran(a, got, ld, *box, threads=threads)
            assert np.array_equal(got, want), (ny, ld, box, threads)


# --------------------------------------------------------------------------- shallow water (S9)
@pytest.mark.parametrize("nx,ny,ld_extra", [(37, 23, 0), (64, 48, 1), (5, 4, 0), (1, 1, 2), (130, 7, 3),
                                            (200, 3, 0)])
def test_sw_step_against_independent_numpy(nx, ny, ld_extra):
    """PARITY UNPINNED by the reference (it has no stencil).  orc_sw_step -- per-point GOcean kernels
    called from loop nests -- against tests/sw_numpy.py, a separately written whole-array evaluation of
    DESIGN.md section 6: bit for bit, on even and odd leading dimensions, full and partial boxes."""
    import sw_numpy as N
    ld, nyarr = nx + 3 + ld_extra, ny + 3
    rng = np.random.default_rng(nx * 1000 + ny)
    prm = N.Params(1.0e5, 0.7e5, 90.0)                     # dx != dy: fsdx/fsdy and tdtsdx/tdtsdy not interchangeable
    u, v, uold, vold = (rng.random((nyarr, ld)) - 0.5 for _ in range(4))
    p, pold = (rng.random((nyarr, ld)) + 1.0 for _ in range(2))
    boxes = [(2, nx + 1, 2, ny + 1)]
    if nx > 8 and ny > 4:
        boxes += [(3, nx - 2, 3, ny), (2, 2, 2, ny + 1), (nx + 1, nx + 1, 4, 4), (5, 4, 2, 3)]
    for box in boxes:
        want = [np.full((nyarr, ld), 9.0) for _ in range(3)]
        got = [np.full((nyarr, ld), 9.0) for _ in range(3)]
        N.sw_step_numpy(prm, box, u, v, p, uold, vold, pold, *want)
        O.sw_step(prm, ld, box, u, v, p, uold, vold, pold, *got)
        for name, g, w in zip(("unew", "vnew", "pnew"), got, want):
            assert np.array_equal(g, w), (name, box)
        xs, xe, ys, ye = box
        if xe >= xs and ye >= ys:
            inner = got[2][ys - 1:ye, xs - 1:xe]
            assert np.all(inner != 9.0) and np.all(np.isfinite(inner))
            got[2][ys - 1:ye, xs - 1:xe] = 9.0
        assert np.all(got[2] == 9.0)                       # nothing outside the box is written


@pytest.mark.parametrize("threads", [1, 3])
def test_fortran_psy_shallow_loops_equal_the_c_oracle(threads):
    """the CPU form a GOcean application runs (pointwise Fortran kernels called from seven PSy loop nests, OpenMP over
    jj; oracle/cpu_psy_loops.f90 -- timed by bench.py's shallow-water cpu_baseline) == orc_sw_step, bit for bit"""
    import sw_numpy as N
    ld, nyarr, box = 70, 41, (2, 68, 2, 39)
    rng = np.random.default_rng(5)
    prm = N.Params(1.0e5, 0.7e5, 90.0)
    u, v, uold, vold = (rng.random((nyarr, ld)) - 0.5 for _ in range(4))
    p, pold = (rng.random((nyarr, ld)) + 1.0 for _ in range(2))
    want = [np.full((nyarr, ld), 9.0) for _ in range(3)]
    got = [np.full((nyarr, ld), 9.0) for _ in range(3)]
    O.sw_step(prm, ld, box, u, v, p, uold, vold, pold, *want)
    O.sw_step_fortran(prm, ld, box, u, v, p, uold, vold, pold, *got, threads=threads)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_sw_constant_state_is_a_fixed_point():
    """u = v = c, p = P everywhere: z = 0, h uniform, divergence 0 => new == old exactly"""
    import sw_numpy as N
    ld, nyarr, box = 40, 30, (2, 38, 2, 28)
    prm = N.Params(1.0e5, 1.0e5, 90.0)
    u = np.full((nyarr, ld), 0.25)
    p = np.full((nyarr, ld), 1.5)
    new = [np.full((nyarr, ld), 9.0) for _ in range(3)]
    O.sw_step(prm, ld, box, u, u, p, u, u, p, *new)
    assert np.all(new[0][1:28, 1:38] == 0.25) and np.all(new[1][1:28, 1:38] == 0.25)
    assert np.all(new[2][1:28, 1:38] == 1.5)


def test_sw_oracle_reproduces_numpy_golden():
    """the committed multi-step golden (tests/golden/make_sw_golden.py, numpy path): 64x48, leapfrog
    rotation, steps 1/5/10 -- every bit of u, v, p (sha256), exact abs-sums and samples"""
    import sw_numpy as N
    gold = load_golden("sw_numpy_64x48")
    nx, ny, ld, nyarr = gold["nx"], gold["ny"], gold["ld"], gold["ny_arr"]
    assert (ld, nyarr) == O.grid_extents(nx + 2, ny + 2)
    box, whole = (2, nx + 1, 2, ny + 1), (1, nx + 2, 1, ny + 2)
    prm = N.Params(gold["dx"], gold["dy"], gold["dt"])
    cur, old, new = N.initial_state(O.hash_field, gold["seed"], nyarr, ld, whole)
    seen = []

    def step(c, o, n):
        O.sw_step(prm, ld, box, *c, *o, *n)

    def on_step(k, c):
        rec = gold["steps"].get(str(k))
        if rec is None:
            return
        seen.append(k)
        for name, f in zip("uvp", c):
            assert N.digest(f, box) == rec[name]["sha256"], (k, name)
            assert N.abs_sum(f, box) == float.fromhex(rec[name]["abs_sum"]), (k, name)
            for (i, j, hx) in rec[name]["samples"]:
                assert f[j - 1, i - 1] == float.fromhex(hx), (k, name, i, j)

    N.leapfrog(step, 10, cur, old, new, on_step)
    assert seen == [1, 5, 10]


# --------------------------------------------------------------------------- T mask (section 8 f.4)
def test_tmask_fill_matches_reference():
    """orc_tmask_fill against grid%tmask of the REAL reference (tests/golden/ref_tmask.json)"""
    for c in load_golden("ref_tmask")["cases"]:
        nx, ny = c["nx"], c["ny"]
        gnx, gny = c["grid"][0], c["grid"][1]
        assert (gnx, gny) == O.grid_extents(nx + 2, ny + 2, c["alignment"])
        user = np.fromfunction(lambda j, i: (7 * (i + 1) + 13 * (j + 1)) % 3 - 1, (ny + 2, nx + 2), dtype=np.int64)
        got = O.tmask_fill(user.astype(np.int32), gnx, gny, (2, nx + 1, 2, ny + 1))
        assert got.tolist() == c["tmask"]
    # no mask supplied: all wet on the subdomain and its ring (grid_mod.f90:447-453)
    m = O.tmask_fill(None, 13, 7, (2, 11, 2, 5))
    assert m[:6, :12].min() == 1 and m.sum() == 6 * 12


def test_jacobi5_masked_against_numpy():
    """PARITY UNPINNED by the reference (it has no stencil).  orc_jacobi5_masked (per-point GOcean
    kernel form) against a whole-array numpy evaluation of DESIGN.md section 5.7, bit for bit"""
    rng = np.random.default_rng(5)
    for n, ld in [(37, 44), (64, 67), (5, 8)]:
        a = rng.random((n + 3, ld))
        tm = rng.integers(-1, 2, (n + 3, ld)).astype(np.int32)
        out = np.full_like(a, -7.0)
        O.jacobi5_masked(a, out, tm, ld, 2, n + 1, 2, n + 1)
        c = a[1:n + 1, 1:n + 1]
        wet = lambda m, v: np.where(m > 0, v, c)                               # noqa: E731
        w = wet(tm[1:n + 1, 0:n], a[1:n + 1, 0:n])
        e = wet(tm[1:n + 1, 2:n + 2], a[1:n + 1, 2:n + 2])
        s_ = wet(tm[0:n, 1:n + 1], a[0:n, 1:n + 1])
        n_ = wet(tm[2:n + 2, 1:n + 1], a[2:n + 2, 1:n + 1])
        want = np.where(tm[1:n + 1, 1:n + 1] > 0, 0.25 * ((w + e) + (s_ + n_)), c)
        assert np.array_equal(out[1:n + 1, 1:n + 1], want)
        out[1:n + 1, 1:n + 1] = -7.0
        assert np.all(out == -7.0)
    # an all-wet mask reduces it to the plain Jacobi step
    a = rng.random((20, 24))
    one = np.ones((20, 24), dtype=np.int32)
    o1, o2 = np.zeros_like(a), np.zeros_like(a)
    O.jacobi5_masked(a, o1, one, 24, 2, 22, 2, 18)
    O.jacobi5(a, o2, 24, 2, 22, 2, 18)
    assert np.array_equal(o1, o2)


# --------------------------------------------------------------------------- SW offset + periodic boundaries
def _periodic_cases():
    return [c for c in load_golden("ref_bounds")["cases"] if not c["abort"] and c["offset"] == 0 and c["halos"]]


def test_periodic_halo_regions_match_reference():
    """orc_periodic_halos against the halo lists the REAL reference built (init_periodic_bc_halos,
    field_mod.f90:1394-1464; tests/golden/ref_bounds.json): every SW-offset periodic case"""
    cases = _periodic_cases()
    assert len(cases) >= 100
    for c in cases:
        it = c["internal"][:4]
        got = O.periodic_halos(it, c["bcx"], c["bcy"])
        assert [s + d for (s, d) in got] == c["halos"], c
        assert len(got) == c["num_halos"]


@pytest.mark.parametrize("nx,ny,ld_extra", [(37, 23, 0), (10, 10, 1), (64, 48, 5), (1, 1, 0), (130, 7, 2)])
def test_sw_offset_step_against_independent_numpy(nx, ny, ld_extra):
    """PARITY UNPINNED by the reference (no stencil there; the SW-offset kernels are the public GOcean
    `shallow` benchmark's).  orc_sw_step_sw against the whole-array numpy
    evaluation sw_step_numpy_sw: bit for bit"""
    import sw_numpy as N
    ld, nyarr = nx + 3 + ld_extra, ny + 3
    rng = np.random.default_rng(nx * 31 + ny)
    prm = N.Params(1.0e5, 0.6e5, 90.0)
    u, v, uold, vold = (rng.random((nyarr, ld)) - 0.5 for _ in range(4))
    p, pold = (rng.random((nyarr, ld)) + 1.0 for _ in range(2))
    boxes = [(2, nx + 1, 2, ny + 1)] + ([(3, nx, 4, ny - 1)] if nx > 8 and ny > 6 else [])
    for box in boxes:
        want = [np.full((nyarr, ld), 9.0) for _ in range(3)]
        got = [np.full((nyarr, ld), 9.0) for _ in range(3)]
        N.sw_step_numpy_sw(prm, box, u, v, p, uold, vold, pold, *want)
        O.sw_step_sw(prm, ld, box, u, v, p, uold, vold, pold, *got)
        for name, g, w in zip(("unew", "vnew", "pnew"), got, want):
            assert np.array_equal(g, w), (name, box)
        xs, xe, ys, ye = box
        assert np.all(np.isfinite(got[0][ys - 1:ye, xs - 1:xe]))


@pytest.mark.parametrize("sw_offset", [False, True], ids=["NE", "SW"])
@pytest.mark.parametrize("name", O.SW_KERNELS)
def test_each_shallow_kernel_against_independent_numpy(name, sw_offset):
    """the GOcean shallow kernels ONE BY ONE (the per-kernel launch entries' checker, orc_sw_kernel: a PSy loop
    nest around each compute_*_code) against the whole-array numpy expressions of tests/sw_numpy.py -- on
    arbitrary input arrays (the intermediates a kernel takes are NOT derived from a state here), even and odd
    leading dimensions, boxes that touch the array edge wherever the kernel's stencil allows it.  Bit for bit.
    PARITY UNPINNED by the reference (no stencil there)."""
    import sw_numpy as N
    prm = N.Params(1.0e5, 0.7e5, 90.0)
    s0, s1 = N.kernel_scalars(name, prm)
    rw, re, rs, rn = N.KERNEL_RING[sw_offset][name]
    for (ld, nyarr) in [(40, 31), (37, 12), (131, 9), (3, 3)]:
        rng = np.random.default_rng(ld * 100 + nyarr + len(name))
        ins = [rng.random((nyarr, ld)) + 0.5 for _ in range(N.KERNEL_NIN[name])]
        tight = (1 + rw, ld - re, 1 + rs, nyarr - rn)                  # the largest box the stencil allows
        boxes = [tight, (2, ld - 1, 2, nyarr - 1), (3, ld - 2, 3, nyarr - 2), (tight[0], tight[0], tight[2], tight[3]),
                 (tight[0], tight[1], tight[3], tight[3]), (5, 4, 2, 3)]
        for box in boxes:
            if box[0] < tight[0] or box[1] > tight[1] or box[2] < tight[2] or box[3] > tight[3]:
                continue
            if name == "time_smooth":
                got, want = ins[2].copy(), ins[2].copy()
                O.sw_kernel(name, sw_offset, ld, box, got, [ins[0], ins[1], got], s0, s1)
                N.kernel_numpy(name, sw_offset, prm, box, want, [ins[0], ins[1], want], alpha=s0)
            else:
                got, want = np.full((nyarr, ld), 9.0), np.full((nyarr, ld), 9.0)
                O.sw_kernel(name, sw_offset, ld, box, got, ins, s0, s1)
                N.kernel_numpy(name, sw_offset, prm, box, want, ins)
            assert np.array_equal(got, want), (name, sw_offset, ld, nyarr, box)
            xs, xe, ys, ye = box
            if xe >= xs and ye >= ys and name != "time_smooth":
                assert np.all(got[ys - 1:ye, xs - 1:xe] != 9.0)
                got[ys - 1:ye, xs - 1:xe] = 9.0
                assert np.all(got == 9.0)                               # nothing outside the box is written


@pytest.mark.parametrize("sw_offset", [False, True], ids=["NE", "SW"])
def test_kernel_sequence_equals_the_fused_oracle_step(sw_offset):
    """the seven loop nests in the order a GOcean PSy layer runs them -- cu, cv, z, h over the box grown towards
    their consumers, then unew, vnew, pnew over the box -- write what orc_sw_step / orc_sw_step_sw write"""
    import sw_numpy as N
    ld, nyarr, box = 45, 33, (2, 43, 2, 31)
    xs, xe, ys, ye = box
    rng = np.random.default_rng(7 + sw_offset)
    prm = N.Params(1.0e5, 0.8e5, 90.0)
    u, v, uold, vold = (rng.random((nyarr, ld)) - 0.5 for _ in range(4))
    p, pold = (rng.random((nyarr, ld)) + 1.0 for _ in range(2))
    want = [np.full((nyarr, ld), 9.0) for _ in range(3)]
    (O.sw_step_sw if sw_offset else O.sw_step)(prm, ld, box, u, v, p, uold, vold, pold, *want)
    cu, cv, z, h = (np.full((nyarr, ld), np.nan) for _ in range(4))
    grown = {False: {"cu": (xs - 1, xe, ys, ye + 1), "cv": (xs, xe + 1, ys - 1, ye), "z": (xs - 1, xe, ys - 1, ye),
                     "h": (xs, xe + 1, ys, ye + 1)},
             True: {"cu": (xs, xe + 1, ys - 1, ye), "cv": (xs - 1, xe, ys, ye + 1), "z": (xs, xe + 1, ys, ye + 1),
                    "h": (xs - 1, xe, ys - 1, ye)}}[sw_offset]
    O.sw_kernel("cu", sw_offset, ld, grown["cu"], cu, [p, u])
    O.sw_kernel("cv", sw_offset, ld, grown["cv"], cv, [p, v])
    O.sw_kernel("z", sw_offset, ld, grown["z"], z, [p, u, v], prm.fsdx, prm.fsdy)
    O.sw_kernel("h", sw_offset, ld, grown["h"], h, [p, u, v])
    got = [np.full((nyarr, ld), 9.0) for _ in range(3)]
    O.sw_kernel("unew", sw_offset, ld, box, got[0], [uold, z, cv, h], prm.tdts8, prm.tdtsdx)
    O.sw_kernel("vnew", sw_offset, ld, box, got[1], [vold, z, cu, h], prm.tdts8, prm.tdtsdy)
    O.sw_kernel("pnew", sw_offset, ld, box, got[2], [pold, cu, cv], prm.tdtsdx, prm.tdtsdy)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_periodic_sw_model_conserves_mass_and_is_translation_invariant():
    """properties of the periodic model (step + halo copies + rotation) that need no reference:
    (i) pnew - pold is a discrete divergence, so SUM(p) over the periodic domain returns to SUM(pold)
    up to rounding; (ii) shifting the initial state by (3, 5) cells around the torus shifts the result"""
    import sw_numpy as N
    n, ld = 24, 27
    prm = N.Params(1.0e5, 1.0e5, 90.0)
    it = (2, n + 1, 2, n + 1)
    rng = np.random.default_rng(11)

    def state(shift):
        rng2 = np.random.default_rng(11)
        out = []
        for k in range(3):
            core = rng2.random((n, n)) + (1.0 if k == 2 else -0.5)
            core = np.roll(core, shift, axis=(0, 1))
            f = np.zeros((n + 3, ld))
            f[1:n + 1, 1:n + 1] = core
            O.apply_periodic_halos(f, ld, it, 0, 0)
            out.append(f)
        return out

    def run(shift, steps=3):
        cur = state(shift)
        old = [f.copy() for f in cur]
        new = [f.copy() for f in cur]
        for _ in range(steps):
            O.sw_step_sw(prm, ld, it, *cur, *old, *new)
            for f in new:
                O.apply_periodic_halos(f, ld, it, 0, 0)
            old, cur, new = cur, new, old
        return cur, old

    cur, old = run((0, 0), steps=1)
    p0 = state((0, 0))[2]
    assert abs(cur[2][1:n + 1, 1:n + 1].sum() - p0[1:n + 1, 1:n + 1].sum()) < 1e-9
    a, _ = run((0, 0))
    b, _ = run((5, 3))
    for fa, fb in zip(a, b):
        assert np.array_equal(np.roll(fa[1:n + 1, 1:n + 1], (5, 3), axis=(0, 1)), fb[1:n + 1, 1:n + 1])


# --------------------------------------------------------------------------- continuity kernel
def test_continuity_against_numpy():
    """PARITY UNPINNED by the reference (it holds no such kernel).  orc_continuity (per-point GOcean
    kernel form) against a whole-array numpy evaluation of DESIGN.md section 5.10, bit for bit; no flow
    leaves the surface where it was; a uniform flow over a flat bed too"""
    rng = np.random.default_rng(10)
    for n, m, ld in [(37, 23, 44), (64, 48, 67), (5, 3, 8), (1, 1, 3)]:
        st, su, sv, hu, hv, un, vn = (rng.random((m + 3, ld)) - 0.3 for _ in range(7))
        area = rng.random((m + 3, ld)) + 0.5
        out = np.full((m + 3, ld), -7.0)
        O.continuity(0.37, ld, (2, n + 1, 2, m + 1), st, su, sv, hu, hv, un, vn, area, out)
        V = lambda a, dj, di: a[1 + dj:m + 1 + dj, 1 + di:n + 1 + di]          # noqa: E731
        r1 = (V(su, 0, 0) + V(hu, 0, 0)) * V(un, 0, 0)
        r2 = (V(su, 0, -1) + V(hu, 0, -1)) * V(un, 0, -1)
        r3 = (V(sv, 0, 0) + V(hv, 0, 0)) * V(vn, 0, 0)
        r4 = (V(sv, -1, 0) + V(hv, -1, 0)) * V(vn, -1, 0)
        want = V(st, 0, 0) + (((r2 - r1) + r4) - r3) * 0.37 / V(area, 0, 0)
        assert np.array_equal(out[1:m + 1, 1:n + 1], want)
        out[1:m + 1, 1:n + 1] = -7.0
        assert np.all(out == -7.0)
        zero = np.zeros_like(un)
        O.continuity(0.37, ld, (2, n + 1, 2, m + 1), st, su, sv, hu, hv, zero, zero, area, out)
        assert np.array_equal(out[1:m + 1, 1:n + 1], st[1:m + 1, 1:n + 1])
        one, two = np.ones_like(un), np.full_like(un, 2.0)
        O.continuity(0.37, ld, (2, n + 1, 2, m + 1), st, zero, zero, two, two, one, one, area, out)
        assert np.array_equal(out[1:m + 1, 1:n + 1], st[1:m + 1, 1:n + 1])          # (2 - 2 + 2 - 2) = 0


# --------------------------------------------------------------------------- general 9-point stencil
def test_stencil9_against_numpy():
    """PARITY UNPINNED by the reference (it has no stencil).  orc_stencil9 (per-point GOcean kernel
    form) against a whole-array numpy evaluation of DESIGN.md section 5.9, bit for bit; a 5-point
    Laplacian and the identity as special cases"""
    rng = np.random.default_rng(9)
    for n, m, ld in [(37, 23, 44), (64, 48, 67), (5, 3, 8), (1, 1, 3)]:
        a = rng.random((m + 3, ld)) - 0.3
        c = rng.random(9) - 0.5
        out = np.full_like(a, -7.0)
        O.stencil9(a, out, c, ld, 2, n + 1, 2, m + 1)
        V = lambda dj, di: a[1 + dj:m + 1 + dj, 1 + di:n + 1 + di]            # noqa: E731
        S = (c[0] * V(-1, -1) + c[1] * V(-1, 0)) + c[2] * V(-1, 1)
        M = (c[3] * V(0, -1) + c[4] * V(0, 0)) + c[5] * V(0, 1)
        N = (c[6] * V(1, -1) + c[7] * V(1, 0)) + c[8] * V(1, 1)
        assert np.array_equal(out[1:m + 1, 1:n + 1], (S + M) + N)
        out[1:m + 1, 1:n + 1] = -7.0
        assert np.all(out == -7.0)
    a = rng.random((20, 24))
    o = np.zeros_like(a)
    O.stencil9(a, o, [0, 0, 0, 0, 1, 0, 0, 0, 0], 24, 2, 22, 2, 18)
    assert np.array_equal(o[1:18, 1:22], a[1:18, 1:22])
    O.stencil9(a, o, [0, 1, 0, 1, -4, 1, 0, 1, 0], 24, 2, 22, 2, 18)
    lap = (a[0:17, 1:22] + a[2:19, 1:22] + a[1:18, 0:21] + a[1:18, 2:23]) - 4 * a[1:18, 1:22]
    assert np.allclose(o[1:18, 1:22], lap, rtol=0, atol=1e-14)
