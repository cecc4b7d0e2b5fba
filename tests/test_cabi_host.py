"""CPU tests of the C-ABI library: it loads, exports every symbol include/dlesm_hip.h declares,
and its host-side index maps (extents, bounds, decomposition, message tables) are bit-exact
with the reference goldens and with the oracle.  No device compute is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import ref_cases as R
from conftest import ROOT, load_golden

import dl_esm_inf_amd as D
from dl_esm_inf_amd import _cabi

L = _cabi.lib()


# --------------------------------------------------------------------------- the boundary
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dlesm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dlesm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 35
    out = subprocess.check_output(["nm", "-D", "--defined-only", _cabi.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    # and the ctypes binding covers exactly the header
    assert sorted(_cabi.PROTOTYPES) == syms
    assert L.dlesm_version() == 310


def test_lab_tool_library_exports_its_header():
    """libdlesm_lab.so (measurement tooling loaded next to the product) == include/dlesm_lab.h == the ctypes table"""
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dlesm_lab.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(dlesm_lab_[a-z0-9_]+)\s*\(", txt)))
    out = subprocess.check_output(["nm", "-D", "--defined-only", _cabi.LAB_LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "dlesm_" in l)
    assert exported == syms == sorted(_cabi.LAB_PROTOTYPES)


def test_product_and_lab_build_export_the_same_entries():
    """the lab build is the same sources with -DDLESM_LAB: same C ABI, more kernels behind it"""
    def entries(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("dlesm_"))
    assert entries(_cabi.LIB_PATH) == entries(_cabi.LAB_BUILD_PATH) == _header_symbols()
    assert os.path.getsize(_cabi.LIB_PATH) < 0.6 * os.path.getsize(_cabi.LAB_BUILD_PATH)      # the variants are the bulk of the code


def test_integration_md_names_every_entry_of_the_product_library():
    """the default .so exports only entries INTEGRATION.md names"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"\bdlesm_[a-z0-9_]+", doc))
    missing = [s_ for s_ in _header_symbols() if s_ not in named]
    assert not missing, missing


def test_tuning_keys_are_classified():
    """every key the sources read is in the library's table (dlesm_tuning_class), the USER keys are the ones INTEGRATION.md
    lists, and tests/conftest.py's rule for "needs the lab build" agrees with the library's LAB class"""
    import conftest
    src = ""
    d = os.path.join(ROOT, "dl_esm_inf_amd", "csrc")
    for f in os.listdir(d):
        if f.endswith((".hip", ".cpp", ".h")):
            src += open(os.path.join(d, f)).read()
    keys = sorted(set(re.findall(r'tuning(?:_nolock)?\("([a-z0-9_]+)"', src)))
    assert len(keys) > 50
    cls = {k: L.dlesm_tuning_class(k.encode()) for k in keys}
    assert not [k for k, c in cls.items() if c < 0], [k for k, c in cls.items() if c < 0]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    settings = doc[doc.index("## Settings"):]
    settings = settings[:settings.index("\n## ", 5)] if "\n## " in settings[5:] else settings
    for k, c in cls.items():
        assert (f"`{k}`" in settings) == (c == 0), (k, c)
    lab = {k for k, c in cls.items() if c == 2}
    assert set(conftest._LAB_DEFAULTS) <= lab, set(conftest._LAB_DEFAULTS) - lab
    for k in lab - {"j5_padw", "util_segp"}:          # (shape-cost weight, segment size: no test selects them)
        assert k in conftest._LAB_DEFAULTS, k


def test_struct_layouts_match_header():
    assert C.sizeof(_cabi.Region) == 24 and C.sizeof(_cabi.Subdomain) == 48
    assert C.sizeof(_cabi.Decomp) == 28
    assert C.sizeof(_cabi.CommTables) == 4 * (2 + 16 * 16)
    assert C.sizeof(_cabi.SwParams) == 40


def test_no_gpu_fails_loudly_not_silently():
    if L.dlesm_device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.zeros((8, 8))
    rc = L.dlesm_stencil5_f64(a.ctypes.data, a.ctypes.data, 8, 8, 2, 7, 2, 7, None)
    assert rc == _cabi.ENODEV
    assert b"no HIP device" in L.dlesm_last_error()
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    D.parallel_init(0, 1)
    g.decompose(8, 8)
    D.grid_init(g, 1.0, 1.0)
    with pytest.raises(D.DlesmError):
        D.r2d_field(g, D.GO_T_POINTS)


def test_product_never_touches_the_oracle():
    """the shipped package must not import, link or open anything under oracle/"""
    pkg = os.path.join(ROOT, "dl_esm_inf_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "lib" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".f90", ".F90", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "dlesm_oracle" not in txt and "oracle_lib" not in txt, \
                    os.path.join(dirpath, f)
    needed = subprocess.check_output(["readelf", "-d", _cabi.LIB_PATH], text=True)
    assert "oracle" not in needed


# --------------------------------------------------------------------------- alignment env
@pytest.mark.parametrize("val,want", [(None, 1), ("1", 1), ("8", 8), ("64", 64), ("128", 128),
                                      (" 16", 16), ("16 ", 16), ("999", 999), ("+4", 4)])
def test_alignment_env_ok(monkeypatch, val, want):
    if val is None:
        monkeypatch.delenv("DL_ESM_ALIGNMENT", raising=False)
    else:
        monkeypatch.setenv("DL_ESM_ALIGNMENT", val)
    a = C.c_int(-1)
    assert L.dlesm_alignment_from_env(C.byref(a)) == 0
    assert a.value == want


@pytest.mark.parametrize("val", ["0", "-8", "abc", "1024", "8x", ""])
def test_alignment_env_aborts_like_reference(monkeypatch, val):
    """grid_mod.f90:353-362: >3 characters, non-numeric or < 1 is fatal"""
    monkeypatch.setenv("DL_ESM_ALIGNMENT", val)
    a = C.c_int(-1)
    assert L.dlesm_alignment_from_env(C.byref(a)) == _cabi.EABORT
    assert b"DL_ESM_ALIGNMENT" in L.dlesm_last_error()


# --------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("case", load_golden("ref_decomp")["cases"],
                         ids=lambda c: f"{c['domainx']}x{c['domainy']}n{c['ndomains']}")
def test_go_decompose_matches_reference(case):
    d = D.go_decompose(case["domainx"], case["domainy"], ndomains=case["ndomains"])
    assert [d.global_nx, d.global_ny, d.nx, d.ny, d.ndomains, d.max_width, d.max_height] == \
           [case["global_nx"], case["global_ny"], case["nx"], case["ny"], case["ndom_out"],
            case["max_width"], case["max_height"]]
    assert [s.glob.as6() + s.internal.as6() for s in d.subdomains] == case["subdomains"]


def test_extents_and_bounds_match_reference(monkeypatch):
    """grid_type/decompose/grid_init/r2d_field bounds through the Python mirror, i.e. through
    the same C-ABI calls the Fortran layer makes"""
    D.parallel_init(0, 1)
    n_abort = 0
    for c in load_golden("ref_bounds")["cases"]:
        if c["alignment"] is None:
            monkeypatch.delenv("DL_ESM_ALIGNMENT", raising=False)
        else:
            monkeypatch.setenv("DL_ESM_ALIGNMENT", str(c["alignment"]))
        g = D.grid_type(D.GO_ARAKAWA_C, (c["bcx"], c["bcy"], D.GO_BC_NONE), c["offset"])
        g.decompose(c["nx"], c["ny"])
        D.grid_init(g, 1.0, 1.0)
        assert [g.nx, g.ny, g.global_nx, g.global_ny] == c["grid"]
        try:
            internal, whole = D.field_mod.field_bounds(g, c["ptype"])
        except D.GoceanStop as e:
            assert c["abort"], (c, str(e))
            n_abort += 1
            continue
        assert not c["abort"], c
        assert internal.as6() == c["internal"] and whole.as6() == c["whole"], c
    assert n_abort == 180


def test_user_tiling_and_bad_arguments():
    d = D.go_decompose(12, 9, ndomainx=3, ndomainy=2)
    assert (d.nx, d.ny, d.ndomains) == (3, 2, 6)
    with pytest.raises(D.GoceanStop):
        D.go_decompose(12, 9, ndomainx=3)                    # parallel_mod.f90:120-122
    info, subs = _cabi.Decomp(), (_cabi.Subdomain * 4)()
    assert L.dlesm_decompose(10, 10, 4, 3, 2, 1, C.byref(info), subs) == _cabi.EINVAL
    assert L.dlesm_decompose(0, 10, 4, 0, 0, 1, C.byref(info), subs) == _cabi.EINVAL


# --------------------------------------------------------------------------- vs the oracle
def _tables_equal(t, c):
    if (t.nsend, t.nrecv) != (c.nsend, c.nrecv):
        return False
    return t.sends() == c.sends() and t.recvs() == c.recvs()


MESHES = R.HALO_CASES + R.GSUM_CASES + [(16, 32, 8), (32, 16, 8), (13, 13, 9), (64, 64, 16),
                                        (100, 37, 7), (37, 100, 7), (12, 9, 5), (7, 5, 2),
                                        (16384, 32768, 8), (31, 29, 12), (9, 50, 10), (50, 9, 3)]


@pytest.mark.parametrize("nx,ny,nranks", MESHES)
def test_map_comms_bit_exact_with_oracle(nx, ny, nranks):
    """the mesh-first table builder must reproduce the reference's border-scan tables, entry for
    entry and in the same order, for every rank"""
    od, osubs = O.decompose(nx, ny, nranks)
    d = D.go_decompose(nx, ny, ndomains=nranks)
    assert [s.glob.as6() + s.internal.as6() for s in d.subdomains] == \
           [s.glob.as6() + s.internal.as6() for s in osubs]
    for r in range(1, nranks + 1):
        t = D.map_comms(d, rank1=r, nranks=nranks)
        c = O.map_comms(od, osubs, nranks, r)
        assert _tables_equal(t, c), (r, t.sends(), c.sends(), t.recvs(), c.recvs())
        # unset slots carry the reference's -999 sentinel
        assert all(t.dirsend[k] == -999 for k in range(t.nsend, 16))
    # iprocmap agrees on a sample of points including outside the domain
    rng = np.random.default_rng(nx * 131 + ny)
    for _ in range(200):
        ia, ja = int(rng.integers(-1, nx + 3)), int(rng.integers(-1, ny + 3))
        assert L.dlesm_iprocmap(C.byref(d._info), d.subdomains, nranks, ia, ja) == \
               O.lib().orc_iprocmap(C.byref(od), osubs, nranks, ia, ja)


def test_map_comms_matches_survey_probe():
    g = load_golden("survey_probe_mapcomms")
    d = D.go_decompose(g["domainx"], g["domainy"], ndomains=g["nranks"])
    t1 = D.map_comms(d, rank1=1, nranks=8)
    for got, want in zip(t1.sends(), g["rank1"]["sends"]):
        assert {k: got[k] for k in want} == want
    for got, want in zip(t1.recvs(), g["rank1"]["recvs"]):
        assert {k: got[k] for k in want} == want
    t4 = D.map_comms(d, rank1=4, nranks=8)
    assert [[s["nx"], s["ny"]] for s in t4.sends()] == g["rank4"]["send_shapes"]


@pytest.mark.parametrize("nx,ny,nranks", [(10, 10, 4), (16, 32, 8), (12, 9, 6), (40, 7, 2), (7, 40, 2)])
def test_map_comms_depth_1_is_the_reference_table_plus_ring_cells(nx, ny, nranks):
    """the depth-d extension at d = 1 on a halo-width-1 decomposition: same messages, in the same
    order, as the reference's tables; a strip that ends at the domain edge is one ring cell longer
    there, and the never-read isrcrecv quirk of the east receive (pcomms:521) is not reproduced"""
    d = D.go_decompose(nx, ny, ndomains=nranks)
    for r in range(1, nranks + 1):
        ref, ext = D.map_comms(d, rank1=r, nranks=nranks), D.map_comms(d, rank1=r, nranks=nranks, depth=1)
        assert [(m["dir"], m["dest"]) for m in ref.sends()] == [(m["dir"], m["dest"]) for m in ext.sends()]
        assert [(m["dir"], m["src"]) for m in ref.recvs()] == [(m["dir"], m["src"]) for m in ext.recvs()]
        for a, b in zip(ref.sends() + ref.recvs(), ext.sends() + ext.recvs()):
            grow_x, grow_y = b["nx"] - a["nx"], b["ny"] - a["ny"]
            assert 0 <= grow_x <= 2 and 0 <= grow_y <= 2 and (a["dir"] <= 4 or grow_x == grow_y == 0)
            assert (a["dir"] in (1, 2) and grow_x == 0) or (a["dir"] in (3, 4) and grow_y == 0) or a["dir"] > 4
            # destination patch starts where the reference's does, or one ring cell before it
            assert a["ides"] - b["ides"] in (0, 1) and a["jdes"] - b["jdes"] in (0, 1)


@pytest.mark.parametrize("nx,ny,nranks,depth", [(16, 32, 8, None), (10, 10, 6, None), (64, 64, 16, None),
                                                (32, 64, 8, 4), (40, 30, 12, 2), (64, 64, 16, 8)])
def test_untagged_message_order_matches_between_every_pair_of_ranks(nx, ny, nranks, depth):
    """RCCL has no tags: between a pair of ranks the k-th send must be the k-th receive.  The plan
    issues sends ordered by (peer, direction code) and receives likewise (dlesm_halo.hip); replay
    that rule on the tables of every rank and compare position by position"""
    d = D.go_decompose(nx, ny, ndomains=nranks, halo_width=depth or 1)
    T = [D.map_comms(d, rank1=r + 1, nranks=nranks, depth=depth) for r in range(nranks)]
    for a in range(nranks):
        for b in range(nranks):
            if a == b:
                continue
            sends = [(m["dir"], m["nx"] * m["ny"]) for m in sorted(T[a].sends(), key=lambda m: (m["dest"], m["dir"]))
                     if m["dest"] == b]
            recvs = [(m["dir"], m["nx"] * m["ny"]) for m in sorted(T[b].recvs(), key=lambda m: (m["src"], m["dir"]))
                     if m["src"] == a]
            assert sends == recvs, (a, b, sends, recvs)


def _calls(t, ld, ny, nf, mask, agg):
    n = C.c_int()
    _cabi.check(L.dlesm_halo_plan_describe(C.byref(t), ld, ny, nf, mask, agg, None, 0, C.byref(n)))
    out = (_cabi.MsgDesc * max(1, n.value))()
    _cabi.check(L.dlesm_halo_plan_describe(C.byref(t), ld, ny, nf, mask, agg, out, n.value, C.byref(n)))
    return [out[k] for k in range(n.value)]


@pytest.mark.parametrize("nx,ny,nranks,depth", [(16, 32, 8, None), (10, 10, 6, None), (64, 64, 16, None), (32, 64, 8, 4)])
@pytest.mark.parametrize("nf,mask,agg", [(1, 0xF, 1), (3, 0xF, 1), (1, 0x1F, 0), (2, 0xF, 0), (2, 0x5, 1), (2, 0xA, 0), (1, 0, 1)])
def test_plan_issue_lists_pair_up_call_by_call(nx, ny, nranks, depth, nf, mask, agg):
    """the same rule on the product's OWN issue lists (dlesm_halo_plan_describe -- what dlesm_halo_exchange*_f64 and the
    distributed steps hand to ncclRecv / ncclSend, in order): for every ordered pair of ranks the k-th send of a to b
    is the k-th receive of b from a (same count, same direction, same field); staging slots of one exchange do not
    overlap, aggregated ones start on 128-byte lines; a masked direction appears on neither side"""
    d = D.go_decompose(nx, ny, ndomains=nranks, halo_width=depth or 1)
    hw = depth or 1
    lists = []
    for r in range(nranks):
        t = D.map_comms(d, rank1=r + 1, nranks=nranks, depth=depth)
        g = d.subdomains[r].glob
        ld, nyy = g.nx + 1 + (r % 2), g.ny + 1             # extents as grid_init makes them (odd and even pitches)
        calls = _calls(t, ld, nyy, nf, mask, agg)
        lists.append(calls)
        seen_send = False
        for c in calls:                                     # per field group: the receives, then the sends
            assert c.peer != r and 0 <= c.peer < nranks and c.count == (nf if c.field < 0 else 1) * c.nx * c.ny
            assert (c.field == -1) == bool(agg)
            seen_send |= not c.is_recv
        for kind in (0, 1):
            spans = sorted((c.buffer_offset, c.buffer_offset + c.count) for c in calls if c.is_recv == kind and c.buffer_offset >= 0)
            assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
            if agg:
                assert all(lo % (16 * nf) == 0 for lo, _ in spans) and len(spans) == sum(c.is_recv == kind for c in calls)
        if not mask & 0xF:
            assert not calls
    for a in range(nranks):
        for b in range(nranks):
            if a == b:
                continue
            sends = [(c.dir, c.field, c.count) for c in lists[a] if not c.is_recv and c.peer == b]
            recvs = [(c.dir, c.field, c.count) for c in lists[b] if c.is_recv and c.peer == a]
            assert sends == recvs, (a, b, sends, recvs)
    assert hw >= 1


def test_map_comms_depth_needs_room():
    t = _cabi.CommTables()
    d = D.go_decompose(40, 40, ndomains=4, halo_width=2)
    assert L.dlesm_map_comms_depth(C.byref(d._info), d.subdomains, 4, 1, 2, C.byref(t)) == 0
    assert {(m["nx"], m["ny"]) for m in t.sends() if m["dir"] > 4} == {(2, 2)}
    assert L.dlesm_map_comms_depth(C.byref(d._info), d.subdomains, 4, 1, 3, C.byref(t)) == _cabi.EINVAL
    assert b"halo width" in L.dlesm_last_error()
    d = D.go_decompose(8, 3, ndomains=4, ndomainx=4, ndomainy=1, halo_width=4)
    assert L.dlesm_map_comms_depth(C.byref(d._info), d.subdomains, 4, 1, 4, C.byref(t)) == _cabi.EINVAL
    assert b"smaller than depth" in L.dlesm_last_error()


def test_map_comms_rejects_what_it_cannot_represent():
    d = D.go_decompose(10, 10, ndomains=4)
    t = _cabi.CommTables()
    assert L.dlesm_map_comms(C.byref(d._info), d.subdomains, 3, 1, C.byref(t)) == _cabi.EINVAL
    assert L.dlesm_map_comms(C.byref(d._info), d.subdomains, 4, 5, C.byref(t)) == _cabi.EINVAL
    d.subdomains[1].glob.xstart += 1            # no longer a tile mesh
    assert L.dlesm_map_comms(C.byref(d._info), d.subdomains, 4, 1, C.byref(t)) == _cabi.EINVAL


def test_grid_extents_random_vs_oracle():
    rng = np.random.default_rng(7)
    for _ in range(500):
        n, m = int(rng.integers(1, 40000)), int(rng.integers(1, 40000))
        a = int(rng.choice([0, 1, 2, 3, 8, 16, 64, 128, 999]))
        nx, ny = C.c_int(), C.c_int()
        assert L.dlesm_grid_extents(n, m, a, C.byref(nx), C.byref(ny)) == 0
        assert (nx.value, ny.value) == O.grid_extents(n, m, a)
        assert nx.value % max(a, 1) == 0 and nx.value > n


# --------------------------------------------------------------------------- id rendezvous (host only)
def test_rendezvous_roundtrip_and_stale_records(tmp_path):
    """dlesm_rendezvous_{remove,publish,fetch}: atomic publish, token and start-time checks"""
    import ctypes as C
    import struct
    import time
    L = D._cabi.lib()
    path = str(tmp_path / "id").encode()
    ident = bytes((7 * k + 3) % 256 for k in range(128))
    got = C.create_string_buffer(128)
    assert L.dlesm_rendezvous_remove(path) == 0                       # nothing there: fine
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run1", 50) == D._cabi.EINVAL
    assert b"no file appeared" in L.dlesm_last_error()
    assert L.dlesm_rendezvous_publish(path, ident, b"4:run1") == 0
    assert os.path.getsize(path) == 256 and not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run1", 50) == 0 and got.raw == ident
    # another job's token, or another world size: ignored
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run2", 50) == D._cabi.EINVAL
    assert b"stale file: job token '4:run1'" in L.dlesm_last_error()
    assert L.dlesm_rendezvous_fetch(path, got, b"8:run1", 50) == D._cabi.EINVAL
    # right token but a publisher that started long before this process: a dead job's leftover
    rec = open(path, "rb").read()
    open(path, "wb").write(rec[:8] + struct.pack("<q", int(time.time()) - 5000) + rec[16:])
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run1", 50) == D._cabi.EINVAL
    assert b"started" in L.dlesm_last_error()
    # a truncated record is not taken either; a fresh publish replaces it
    open(path, "wb").write(rec[:100])
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run1", 50) == D._cabi.EINVAL
    assert L.dlesm_rendezvous_publish(path, ident[::-1], b"4:run1") == 0
    assert L.dlesm_rendezvous_fetch(path, got, b"4:run1", 50) == 0 and got.raw == ident[::-1]
    assert L.dlesm_rendezvous_publish(path, ident, ("x" * 112).encode()) == D._cabi.EINVAL   # token too long
    assert L.dlesm_rendezvous_remove(path) == 0 and not os.path.exists(path)


def test_python_grid_tmask_matches_reference():
    """grid_init(tmask=...) of the Python mirror builds the same grid%tmask as the real reference"""
    for c in load_golden("ref_tmask")["cases"]:
        if c["alignment"]:
            os.environ["DL_ESM_ALIGNMENT"] = str(c["alignment"])
        else:
            os.environ.pop("DL_ESM_ALIGNMENT", None)
        D.parallel_init(0, 1, use_rccl=False)
        g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
        g.decompose(c["nx"], c["ny"])
        user = np.fromfunction(lambda j, i: (7 * (i + 1) + 13 * (j + 1)) % 3 - 1, (c["ny"] + 2, c["nx"] + 2),
                               dtype=np.int64)
        D.grid_init(g, 1.0, 1.0, tmask=user)
        assert [g.nx, g.ny] == c["grid"][:2]
        assert g.tmask.tolist() == c["tmask"]
    os.environ.pop("DL_ESM_ALIGNMENT", None)


def test_periodic_halo_regions_match_reference_and_python_fields_carry_them():
    """dlesm_periodic_halos == the real reference's field%halo lists (ref_bounds.json)"""
    import ctypes as C
    Reg = _cabi.Region
    n_checked = 0
    for c in load_golden("ref_bounds")["cases"]:
        if c["abort"] or c["offset"] != 0 or not c["halos"]:
            continue
        it = Reg(0, 0, *c["internal"][:4])
        src, dst, n = (Reg * 4)(), (Reg * 4)(), C.c_int()
        assert L.dlesm_periodic_halos(C.byref(it), c["bcx"], c["bcy"], src, dst, C.byref(n)) == 0
        got = [[src[k].xstart, src[k].xstop, src[k].ystart, src[k].ystop, dst[k].xstart, dst[k].xstop,
                dst[k].ystart, dst[k].ystop] for k in range(n.value)]
        assert got == c["halos"], c
        assert all(src[k].nx == dst[k].nx and src[k].ny == dst[k].ny for k in range(n.value))
        n_checked += 1
    assert n_checked >= 100


@pytest.mark.parametrize("n", [1, 2, 5])
def test_board_allgather_between_processes(tmp_path, n):
    """the host-side all-gather of mailbox mode (dlesm_board_*, csrc/dlesm_rendezvous.cpp) between real processes: nine
    operations of 0 to 100 000 bytes with the ranks out of step, one of them a gather (only the root reads), then the
    close; every file of the session is gone afterwards.  No GPU, no RCCL."""
    import subprocess
    import sys
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "board_worker.py"), f"pytest-{os.getpid()}-{n}", str(r),
                               str(n), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(n)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0 and f"board rank {r} ok" in out, out[-2000:]
    assert os.listdir(tmp_path) == [], os.listdir(tmp_path)


def test_board_refusals(tmp_path):
    L = _cabi.lib()
    bad = C.create_string_buffer(b"../etc", _cabi.UNIQUE_ID_BYTES)
    assert L.dlesm_board_open(bad, 2, 0) != 0 and "letters, digits" in _cabi.last_error()
    assert L.dlesm_board_allgather(None, 0, None) != 0 and "not open" in _cabi.last_error()
    nonce = C.create_string_buffer(_cabi.UNIQUE_ID_BYTES)
    _cabi.check(L.dlesm_board_nonce(nonce))
    assert nonce.value.startswith(b"mbx-") and L.dlesm_board_is_open() == 0


def test_board_abort_note_ends_the_wait(tmp_path):
    """a rank that stops leaves a note (dlesm_board_abort, what parallel_abort calls in mailbox mode): a rank waiting for it
    on the board fails within a moment with the note's text, not after the time-out"""
    import subprocess
    import sys
    import time
    code = (
        "import ctypes as C, os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['DLESM_BOARD_DIR'] = %r; os.environ['DLESM_BOARD_TIMEOUT_S'] = '60'\n"
        "from dl_esm_inf_amd import _cabi\n"
        "L = _cabi.lib(); sid = C.create_string_buffer(b'abort-test', _cabi.UNIQUE_ID_BYTES)\n"
        "rank = int(sys.argv[1]); _cabi.check(L.dlesm_board_open(sid, 2, rank))\n"
        "if rank == 1:\n"
        "    time.sleep(0.3); L.dlesm_board_abort(b'grid_init: ERROR: something fatal'); sys.exit(1)\n"
        "t0 = time.time(); b = C.create_string_buffer(8); a = C.create_string_buffer(16)\n"
        "rc = L.dlesm_board_allgather(b, 8, a)\n"
        "print('rc', rc, 'waited %%.1f' %% (time.time() - t0), _cabi.last_error())\n"
    ) % (ROOT, str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    out0, _ = procs[0].communicate(timeout=60)
    procs[1].communicate(timeout=60)
    assert f"rc {_cabi.EABORT}" in out0 and "rank 1: grid_init: ERROR: something fatal" in out0, out0
    assert float(out0.split("waited")[1].split()[0]) < 10.0, out0



@pytest.mark.parametrize("nx,ny,nranks,depth", [(16, 32, 8, None), (10, 10, 6, None), (64, 64, 16, None), (32, 64, 8, 4), (13, 11, 9, None),
                                                (7, 40, 5, None), (10, 4, 2, None)])
@pytest.mark.parametrize("nf", [1, 3])
def test_mailbox_matching_on_whole_meshes(nx, ny, nranks, depth, nf):
    """the peer transport's matching -- which receive slot of which neighbour every send strip is stored into -- run by the
    product's own code (dlesm_peer_blob_describe / dlesm_peer_match_describe: the functions dlesm_halo_plan_peer_export and
    _connect call) for EVERY rank of a mesh, BASELINE configs[4]'s 2 x 4 among them, with no device.  Checked against the
    geometry, not against the rule: the strip a send reads, shifted into the neighbour's frame of reference, must be
    exactly the halo strip the matched receive writes there; every receive slot is matched by exactly one send; slots of
    one mailbox do not overlap."""
    d = D.go_decompose(nx, ny, ndomains=nranks, halo_width=depth or 1)
    tabs, dims, blobs = [], [], b""
    for r in range(nranks):
        t = D.map_comms(d, rank1=r + 1, nranks=nranks, depth=depth)
        g = d.subdomains[r].glob
        ld, nyy = g.nx + 1 + (r % 2), g.ny + 1
        blob = C.create_string_buffer(_cabi.PEER_BLOB_BYTES)
        _cabi.check(L.dlesm_peer_blob_describe(C.byref(t), ld, nyy, r, nf, blob))
        tabs.append(t), dims.append((ld, nyy))
        blobs += blob.raw
    allb = C.create_string_buffer(blobs, nranks * _cabi.PEER_BLOB_BYTES)
    # the receives of every rank in the plan's own (peer, direction) order: (source rank, dir, 1-based dest strip)
    recvs = []
    for r in range(nranks):
        calls = [c for c in _calls(tabs[r], *dims[r], 1, 0x1F & 0xF, 1) if c.is_recv]
        recvs.append(calls)
    taken = [set() for _ in range(nranks)]
    for r in range(nranks):
        n = C.c_int()
        out = (_cabi.PeerMatchDesc * 16)()
        _cabi.check(L.dlesm_peer_match_describe(C.byref(tabs[r]), *dims[r], r, nranks, nf, allb, out, 16, C.byref(n)))
        assert n.value == tabs[r].nsend
        for k in range(n.value):
            m = out[k]
            q = m.peer
            rc = recvs[q][m.slot]                       # the receive this send is stored into
            assert rc.peer == r and rc.count == m.count * 1 and (rc.nx, rc.ny) == (m.nx, m.ny), (r, k, q, m.slot)
            # geometry: global coordinates of the cells sent == global coordinates of the halo cells received
            sx = d.subdomains[r].glob.xstart - d.subdomains[r].internal.xstart      # local -> global shift of the sender
            sy = d.subdomains[r].glob.ystart - d.subdomains[r].internal.ystart
            qx = d.subdomains[q].glob.xstart - d.subdomains[q].internal.xstart
            qy = d.subdomains[q].glob.ystart - d.subdomains[q].internal.ystart
            assert (m.i0 + sx, m.j0 + sy) == (rc.i0 + qx, rc.j0 + qy), (r, k, q, (m.i0, m.j0), (rc.i0, rc.j0))
            assert m.slot not in taken[q]
            taken[q].add(m.slot)
            assert m.off * nf == rc.buffer_offset * nf // 1     # the per-field slot offset the neighbour's plan reports (aggregated layout, 1 field)
    for q in range(nranks):
        assert taken[q] == set(range(len(recvs[q]))), (q, taken[q])
    # a blob of the wrong rank in a slot is refused
    if nranks > 1:
        wrong = C.create_string_buffer(blobs[_cabi.PEER_BLOB_BYTES:2 * _cabi.PEER_BLOB_BYTES] + blobs[_cabi.PEER_BLOB_BYTES:],
                                       nranks * _cabi.PEER_BLOB_BYTES)
        n = C.c_int()
        out = (_cabi.PeerMatchDesc * 16)()
        bad = [r for r in range(nranks) if any(tabs[r].destination[k] == 0 for k in range(tabs[r].nsend))]
        if bad:
            assert L.dlesm_peer_match_describe(C.byref(tabs[bad[0]]), *dims[bad[0]], bad[0], nranks, nf, wrong, out, 16, C.byref(n)) != 0


def _mesh_or_none(nx, ny, nranks, depth=None):
    try:
        d = D.go_decompose(nx, ny, ndomains=nranks, halo_width=depth or 1)
        tabs = [D.map_comms(d, rank1=r + 1, nranks=nranks, depth=depth) for r in range(nranks)]
    except Exception:                                        # noqa: BLE001  (domains too small for that many ranks etc.)
        return None
    return d, tabs


def test_mailbox_matching_on_random_meshes():
    """the same geometric check as test_mailbox_matching_on_whole_meshes on a few hundred random (domain, rank count)
    combinations that go_decompose / map_comms accept: 2 to 16 ranks, uneven tiles, thin domains"""
    import random
    rng = random.Random(20261004)
    checked = 0
    for _ in range(400):
        nranks = rng.randint(2, 16)
        nx, ny = rng.randint(2, 90), rng.randint(2, 90)
        mesh = _mesh_or_none(nx, ny, nranks)
        if mesh is None:
            continue
        d, tabs = mesh
        if any(t.nsend > 8 or t.nrecv > 8 for t in tabs):
            continue
        dims, blobs = [], b""
        ok = True
        for r in range(nranks):
            g = d.subdomains[r].glob
            ld, nyy = g.nx + 1 + (r % 3), g.ny + 1
            blob = C.create_string_buffer(_cabi.PEER_BLOB_BYTES)
            if L.dlesm_peer_blob_describe(C.byref(tabs[r]), ld, nyy, r, 2, blob) != 0:
                ok = False
                break
            dims.append((ld, nyy))
            blobs += blob.raw
        if not ok:
            continue
        allb = C.create_string_buffer(blobs, nranks * _cabi.PEER_BLOB_BYTES)
        recvs = [[c for c in _calls(tabs[r], *dims[r], 1, 0xF, 1) if c.is_recv] for r in range(nranks)]
        taken = [set() for _ in range(nranks)]
        for r in range(nranks):
            n = C.c_int()
            out = (_cabi.PeerMatchDesc * 16)()
            _cabi.check(L.dlesm_peer_match_describe(C.byref(tabs[r]), *dims[r], r, nranks, 2, allb, out, 16, C.byref(n)))
            assert n.value == tabs[r].nsend
            sx = d.subdomains[r].glob.xstart - d.subdomains[r].internal.xstart
            sy = d.subdomains[r].glob.ystart - d.subdomains[r].internal.ystart
            for k in range(n.value):
                m = out[k]
                rc = recvs[m.peer][m.slot]
                qx = d.subdomains[m.peer].glob.xstart - d.subdomains[m.peer].internal.xstart
                qy = d.subdomains[m.peer].glob.ystart - d.subdomains[m.peer].internal.ystart
                assert rc.peer == r and (rc.nx, rc.ny) == (m.nx, m.ny) and (m.i0 + sx, m.j0 + sy) == (rc.i0 + qx, rc.j0 + qy), \
                    (nx, ny, nranks, r, k)
                assert m.slot not in taken[m.peer]
                taken[m.peer].add(m.slot)
        for q in range(nranks):
            assert taken[q] == set(range(len(recvs[q]))), (nx, ny, nranks, q)
        checked += 1
    assert checked >= 150, checked
