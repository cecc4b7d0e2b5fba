"""The Fortran API layer (dl_esm_inf_amd/fortran -> lib_fd_hip.a), driven through Fortran
programs exactly as a GOcean application would use it.

CPU part: tests/fortran/ftest_dump.f90 prints the same "G:" lines for this library as
oracle/ref_drivers/ref_dump.f90 printed for the REAL reference; they are compared with the
committed goldens (bit-exact integers) and, for the per-rank message tables, with the oracle.
GPU part (-m gpu): tests/fortran/ftest_device.f90 replays the reference's device-io test on the
real device and runs a Jacobi model through the PSy layer.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import ref_cases as R
from conftest import ROOT, load_golden

FDIR = os.path.join(ROOT, "dl_esm_inf_amd", "fortran")
BUILD = os.path.join(FDIR, "build")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)

    def run(name, *args, env=None, check=True):
        e = dict(os.environ)
        e.pop("DL_ESM_ALIGNMENT", None)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            e.pop(k, None)
        e.update(env or {})
        p = subprocess.run([os.path.join(BUILD, name), *map(str, args)], env=e, capture_output=True,
                           text=True, timeout=300)
        if check:
            assert p.returncode == 0, p.stderr[-2000:]
        g = {}
        for line in p.stdout.splitlines():
            if line.startswith("G: "):
                t = line[3:].split()
                g.setdefault(t[0], []).append(t[1:])
        return p.returncode, g, p.stderr
    return run


def ints(t):
    return [int(x) for x in t]


def test_fortran_go_decompose_matches_reference(exe):
    for c in load_golden("ref_decomp")["cases"]:
        _, g, _ = exe("ftest_dump.exe", "decomp", c["domainx"], c["domainy"], c["ndomains"])
        assert ints(g["decomp"][0]) == [c["global_nx"], c["global_ny"], c["nx"], c["ny"], c["ndom_out"],
                                        c["max_width"], c["max_height"]]
        assert [ints(s)[1:] for s in g["sub"]] == c["subdomains"]


def test_fortran_bounds_match_reference(exe):
    """grid_type + decompose + grid_init + r2d_field through the Fortran layer, incl. the
    combinations on which the reference stops (gocean_stop -> non-zero exit)"""
    cases = [c for c in load_golden("ref_bounds")["cases"]
             if c["alignment"] in (None, 64) and (c["nx"], c["ny"]) in ((10, 4), (256, 256), (5, 5))]
    assert len(cases) == 210
    n_abort = 0
    for c in cases:
        env = {"DL_ESM_ALIGNMENT": str(c["alignment"])} if c["alignment"] else None
        rc, g, err = exe("ftest_dump.exe", "bounds", c["nx"], c["ny"], c["offset"], c["bcx"], c["bcy"],
                         c["ptype"], env=env, check=False)
        assert ints(g["grid"][0]) == c["grid"], c
        if c["abort"]:
            n_abort += 1
            assert rc != 0 and "field" not in g, c
            continue
        assert rc == 0, (c, err)
        f = ints(g["field"][0])
        assert f[0] == c["defined_on"] and f[1:7] == c["internal"] and f[7:13] == c["whole"], c
        if c["num_halos"] is not None:
            assert f[13] == c["num_halos"]
        assert ints(g["shape"][0]) == c["shape"]
        assert [ints(h)[1:] for h in g.get("halo", [])] == c["halos"]
    assert n_abort == 54


def test_fortran_model_and_gather_match_reference(exe):
    gold = load_golden("ref_model")
    for m in gold["model"]:
        _, g, _ = exe("ftest_dump.exe", "model", m["nx"], m["ny"], m["fill"])
        assert ints(g["grid"][0]) == m["grid"] and ints(g["internal"][0]) == m["internal"]
        assert float(g["checksum"][0][0]) == m["checksum"]
        assert [float(x) for x in g["xt"][0]] == m["xt"] and [float(x) for x in g["yt"][0]] == m["yt"]
    for m in gold["gather"]:
        _, g, _ = exe("ftest_dump.exe", "gather", m["nx"], m["ny"])
        assert [float(x) for x in g["corner"][0]] == m["corner"]
        assert float(g["checksum"][0][0]) == m["checksum"]
        assert ints(g["gather_shape"][0]) == m["gather_shape"] and int(g["gather_mismatch"][0][0]) == 0


@pytest.mark.parametrize("nx,ny,nranks", R.HALO_CASES + [(16, 32, 8), (13, 13, 9)])
def test_fortran_message_tables_match_oracle(exe, nx, ny, nranks):
    """every rank's parallel_comms_mod tables after grid_init (RANK/WORLD_SIZE from the
    environment, dry communicator: no GPU) against the oracle's map_comms"""
    od, osubs = O.decompose(nx, ny, nranks)
    for r in range(nranks):
        _, g, _ = exe("ftest_dump.exe", "comms", nx, ny,
                      env={"RANK": str(r), "WORLD_SIZE": str(nranks), "DLESM_DRY_COMMS": "1"})
        c = O.map_comms(od, osubs, nranks, r + 1)
        ext = O.grid_extents(osubs[r].glob.nx, osubs[r].glob.ny)
        assert ints(g["rank"][0]) == [r + 1, nranks, ext[0], ext[1]]
        assert ints(g["counts"][0]) == [c.nsend, c.nrecv]
        want_s = [[s["dir"], s["dest"], s["isrc"], s["jsrc"], s["ides"], s["jdes"], s["nx"], s["ny"]]
                  for s in c.sends()]
        want_r = [[q["dir"], q["src"], q["ides"], q["jdes"], q["nx"], q["ny"]] for q in c.recvs()]
        assert [ints(s) for s in g.get("send", [])] == want_s
        assert [ints(q) for q in g.get("recv", [])] == want_r
        sg = osubs[r].glob
        assert ints(g["bounds"][0]) == [sg.xstart, sg.xstop, sg.ystart, sg.ystop]


@pytest.mark.parametrize("nx,ny,nranks,hw", [(16, 32, 8, 4), (24, 20, 6, 3), (40, 12, 2, 8)])
def test_fortran_deep_halo_tables_match_the_c_abi(exe, nx, ny, nranks, hw):
    """a decomposition made with halo_width > 1 gets the depth-hw tables (the extension of
    dlesm_map_comms_depth) from the Fortran grid_init too"""
    import dl_esm_inf_amd as D
    d = D.go_decompose(nx, ny, ndomains=nranks, halo_width=hw)
    for r in range(nranks):
        _, g, _ = exe("ftest_dump.exe", "comms", nx, ny, hw,
                      env={"RANK": str(r), "WORLD_SIZE": str(nranks), "DLESM_DRY_COMMS": "1"})
        c = D.map_comms(d, rank1=r + 1, nranks=nranks, depth=hw)
        assert ints(g["counts"][0]) == [c.nsend, c.nrecv]
        assert [ints(s) for s in g.get("send", [])] == \
            [[s["dir"], s["dest"], s["isrc"], s["jsrc"], s["ides"], s["jdes"], s["nx"], s["ny"]] for s in c.sends()]
        assert [ints(q) for q in g.get("recv", [])] == \
            [[q["dir"], q["src"], q["ides"], q["jdes"], q["nx"], q["ny"]] for q in c.recvs()]
        assert any(s["nx"] == hw or s["ny"] == hw for s in c.sends())


def test_fortran_tmask_matches_reference(exe):
    """grid_init(tmask=pattern) through the Fortran layer: grid%tmask equals the real reference's
    (tests/golden/ref_tmask.json, oracle/_ref/ref_dump.exe tmask) entry for entry"""
    for c in load_golden("ref_tmask")["cases"]:
        env = {"DL_ESM_ALIGNMENT": str(c["alignment"])} if c["alignment"] else None
        _, g, _ = exe("ftest_dump.exe", "tmask", c["nx"], c["ny"], env=env)
        assert ints(g["grid"][0]) == c["grid"]
        rows = sorted((ints(r) for r in g["tmaskrow"]), key=lambda r: r[0])
        assert [r[1:] for r in rows] == c["tmask"]


def _plant_stale(path, token, age_s):
    """a rendezvous record as a dead job would have left it (format: dlesm_rendezvous.cpp)"""
    import struct
    import time
    rec = b"DLESMRV1" + struct.pack("<q", int(time.time()) - age_s) + token.encode().ljust(112, b"\0") + \
        bytes(range(128))
    with open(path, "wb") as f:
        f.write(rec)


@pytest.mark.parametrize("stale", ["none", "other_job_token", "old_publisher", "garbage"])
def test_fortran_bootstrap_rendezvous_ignores_stale_files(exe, tmp_path, stale):
    """parallel_init's id rendezvous with two real processes (DLESM_DRY_COMMS=2: blank id, no
    communicator, no GPU): rank 0 replaces whatever it finds, a reader that starts FIRST and sees a
    stale record of a dead job (other token / a publisher that started long ago / not a record)
    keeps waiting for this job's record instead of taking the old id; both leave cleanly and the
    file is removed at parallel_finalise"""
    import time
    path = str(tmp_path / "rv")
    env = {"WORLD_SIZE": "2", "DLESM_DRY_COMMS": "2", "DLESM_RENDEZVOUS": path, "DLESM_JOB_ID": "job-B",
           "DLESM_RENDEZVOUS_TIMEOUT_S": "20"}
    if stale == "other_job_token":
        _plant_stale(path, "2:job-A", 0)
    elif stale == "old_publisher":
        _plant_stale(path, "2:job-B", 100000)
    elif stale == "garbage":
        open(path, "wb").write(b"x" * 300)
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "DL_ESM_ALIGNMENT"):
        e.pop(k, None)
    e.update(env)
    cmd = [os.path.join(BUILD, "ftest_dump.exe"), "comms", "10", "4"]
    reader = subprocess.Popen(cmd, env={**e, "RANK": "1"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(0.5)                                     # the reader is polling the (possibly stale) file by now
    assert reader.poll() is None, reader.stderr.read()
    root = subprocess.run(cmd, env={**e, "RANK": "0"}, capture_output=True, text=True, timeout=60)
    out, err = reader.communicate(timeout=60)
    assert root.returncode == 0, root.stderr[-1000:]
    assert reader.returncode == 0, err[-1000:]
    assert "G: rank 2 2" in out and "G: rank 1 2" in root.stdout
    assert not os.path.exists(path)                     # rank 0 removed it on the way out


def test_fortran_bootstrap_times_out_with_a_reason(exe, tmp_path):
    """no rank 0 ever shows up: the reader stops with the reason, it does not hang"""
    path = str(tmp_path / "rv")
    _plant_stale(path, "2:job-A", 0)
    rc, _, err = exe("ftest_dump.exe", "comms", 10, 4, check=False,
                     env={"RANK": "1", "WORLD_SIZE": "2", "DLESM_DRY_COMMS": "2", "DLESM_RENDEZVOUS": path,
                          "DLESM_JOB_ID": "job-B", "DLESM_RENDEZVOUS_TIMEOUT_S": "1"})
    assert rc != 0
    flat = err.replace("\n ", "")                       # list-directed output wraps long lines
    assert "no usable RCCL id" in flat and "stale file" in flat and "job-A" in flat


@pytest.mark.gpu
def test_fortran_device_io_and_jacobi_on_gpu(exe):
    nx, ny, nsteps = 300, 171, 5
    _, g, _ = exe("ftest_device.exe", nx, ny, nsteps, env={"DL_ESM_ALIGNMENT": "8"})
    # (1) the reference's device-io scenario, on the real device
    gold = next(r for r in load_golden("ref_device_io")["runs"] if r["alignment"] == 8)
    assert [[float(x) for x in row] for row in g["io"]] == gold["rows"]
    # (1b) grid-property device mirrors (grid%*_device) hold the host arrays
    assert int(g["mirrors"][0][0]) == 0
    # (2) Jacobi through the Fortran PSy layer == oracle
    ld, nyy = ints(g["grid"][0])
    assert (ld, nyy) == O.grid_extents(nx + 2, ny + 2, 8)
    a = O.hash_field(20261004, nyy, ld, 0, 0, 1, nx + 2, 1, ny + 2)
    cs0 = O.lib().orc_checksum(a, ld, 2, nx + 1, 2, ny + 1)
    assert abs(float(g["cs0"][0][0]) - cs0) <= 1e-12 * cs0
    b = a.copy()
    for _ in range(nsteps):
        O.jacobi5(a, b, ld, 2, nx + 1, 2, ny + 1)
        a, b = b, a
    cs = O.lib().orc_checksum(a, ld, 2, nx + 1, 2, ny + 1)
    assert abs(float(g["cs"][0][0]) - cs) <= 1e-12 * cs
    got = [float(x) for x in g["sample"][0]]
    assert got == [a[1, 1], a[ny // 2, nx // 2], a[ny, nx]]
    assert ints(g["gather"][0]) == [nx, ny, 0]        # device-side gather_inner_data == the field's interior
    # (2b) invoke_jacobi5_multi(.., 4) == four invoke_jacobi5 calls, and == four oracle steps
    assert int(g["fused4"][0][0]) == 0
    a = O.hash_field(4242, nyy, ld, 0, 0, 1, nx + 2, 1, ny + 2)
    b = a.copy()
    for _ in range(4):
        O.jacobi5(a, b, ld, 2, nx + 1, 2, ny + 1)
        a, b = b, a
    cs4 = O.lib().orc_checksum(a, ld, 2, nx + 1, 2, ny + 1)
    assert abs(float(g["fused4"][0][1]) - cs4) <= 1e-12 * cs4
    # (2c) masked Jacobi through the PSy layer (kernel argument GO_GRID_MASK_T -> grid%tmask_device)
    user = np.fromfunction(lambda j, i: (7 * (i + 1) + 13 * (j + 1)) % 3 - 1, (ny + 2, nx + 2), dtype=np.int64)
    tm = O.tmask_fill(user.astype(np.int32), ld, nyy, (2, nx + 1, 2, ny + 1))
    a = O.hash_field(777, nyy, ld, 0, 0, 1, nx + 2, 1, ny + 2)
    b = a.copy()
    for _ in range(nsteps):
        O.jacobi5_masked(a, b, tm, ld, 2, nx + 1, 2, ny + 1)
        a, b = b, a
    csm = O.lib().orc_checksum(a, ld, 2, nx + 1, 2, ny + 1)
    row = [float(x) for x in g["masked"][0]]
    assert abs(row[0] - csm) <= 1e-12 * csm
    assert row[1:] == [a[1, 1], a[ny // 2, nx // 2], a[ny, nx]]
    # (2d) the general 3x3 stencil through the Fortran PSy wrapper (coef(-1:1,-1:1) -> C order)
    a9 = O.hash_field(909, nyy, ld, 0, 0, 1, nx + 2, 1, ny + 2)
    b9 = a9.copy()
    coef = [0.05 * (k + 1) - 0.2 for k in range(9)]
    O.stencil9(a9, b9, coef, ld, 2, nx + 1, 2, ny + 1)
    cs9 = O.lib().orc_checksum(b9, ld, 2, nx + 1, 2, ny + 1)
    row = [float(x) for x in g["s9"][0]]
    assert abs(row[0] - cs9) <= 1e-12 * cs9 and row[1:] == [b9[1, 1], b9[ny, nx]]
    # (3b) two leapfrog steps per launch through the Fortran wrappers == two single-step calls (plain and filtered forms)
    assert ints(g["x2"][0]) == [0, 0]
    # (3) one fused shallow-water step launched from Fortran == oracle
    import ctypes as C
    H = []
    for k in range(1, 7):
        f = O.hash_field(k, nyy, ld, 0, 0, 1, nx + 2, 1, ny + 2)
        f[:ny + 2, :nx + 2] += 1.0 if k % 3 == 0 else -0.5
        H.append(f)
    outs = [np.full((nyy, ld), 9.0) for _ in range(3)]
    scratch = [np.zeros((nyy, ld)) for _ in range(4)]
    tdt = 180.0
    prm = O.SwParams(4.0 / 1.0e5, 4.0 / 1.0e5, tdt / 8.0, tdt / 1.0e5, tdt / 1.0e5)
    O.lib().orc_sw_step(C.byref(prm), ld, 2, nx + 1, 2, ny + 1, *H, *scratch, *outs)
    for row, want in zip(g["sw"], outs):
        cs_w = O.lib().orc_checksum(want, ld, 2, nx + 1, 2, ny + 1)
        assert abs(float(row[1]) - cs_w) <= 1e-12 * cs_w
        assert [float(row[2]), float(row[3])] == [want[1, 1], want[ny, nx]]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4], ids=["fused", "seven-kernels", "seven-kernels+time_smooth", "fused+time_smooth",
                                                      "two-filtered-steps-per-launch"])
@pytest.mark.parametrize("n,nsteps,alignment", [(10, 5, None), (256, 6, 64)])      # (5 steps: mode 4 ends with one single step)
def test_fortran_periodic_shallow_app_on_gpu(exe, n, nsteps, alignment, mode):
    """examples/shallow_app.f90 -- a GOcean-style SW-offset, doubly periodic shallow-water model written
    against the reference's API (grid_type, r2d_field, halo lists) with the PSy layer of this library: the fused
    one-launch step, the seven GOcean kernels launched one by one (what an unmodified generated PSy layer does),
    and the latter with the time_smooth kernel -- against the oracle running the same model: checksums within
    1e-12, sampled cells bit for bit"""
    import sw_numpy as N
    env = {"DL_ESM_ALIGNMENT": str(alignment)} if alignment else None
    _, g, _ = exe("shallow_app.exe", n, nsteps, mode, env=env)
    ld, nyy, _ = ints(g["shape"][0])
    assert (ld, nyy) == O.grid_extents(n + 2, n + 2, alignment)
    it = (2, n + 1, 2, n + 1)
    prm = N.Params(1.0e5, 1.0e5, 90.0)
    cur = []
    for k in range(1, 4):
        f = O.hash_field(100 + k, nyy, ld, 0, 0, *it) + (1.0 if k == 3 else -0.5)
        O.apply_periodic_halos(f, ld, it, 0, 0)
        cur.append(f)
    old, new = [f.copy() for f in cur], [f.copy() for f in cur]
    for _ in range(nsteps):
        O.sw_step_sw(prm, ld, it, *cur, *old, *new)
        for f in new:
            O.apply_periodic_halos(f, ld, it, 0, 0)
        if mode >= 2:
            for c, nw, o in zip(cur, new, old):
                O.sw_kernel("time_smooth", True, ld, it, o, [c, nw, o], 0.001)
                O.apply_periodic_halos(o, ld, it, 0, 0)
            cur, new = new, cur
        else:
            old, cur, new = cur, new, old
    for row, want in zip(g["cs"], cur):
        cs = O.lib().orc_checksum(want, ld, *it)
        assert abs(float(row[1]) - cs) <= 1e-12 * cs
        assert [float(row[2]), float(row[3])] == [want[1, 1], want[n, n]]
