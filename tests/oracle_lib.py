"""ctypes view of oracle/liboracle.so -- the CPU restatement used as the CHECKER.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing in dl_esm_inf_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MAXCOMM = 16


class Region(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "ny", "xstart", "xstop", "ystart", "ystop")]

    def as6(self):
        """xstart,xstop,ystart,ystop,nx,ny -- the order the golden files use"""
        return [self.xstart, self.xstop, self.ystart, self.ystop, self.nx, self.ny]


class Subdomain(C.Structure):
    _fields_ = [("glob", Region), ("internal", Region)]


class Decomp(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("global_nx", "global_ny", "nx", "ny", "ndomains",
                                       "max_width", "max_height")]


_COMM_ARRAYS = ("dirsend", "destination", "isrcsend", "jsrcsend", "idessend", "jdessend",
                "nxsend", "nysend", "dirrecv", "source", "isrcrecv", "jsrcrecv",
                "idesrecv", "jdesrecv", "nxrecv", "nyrecv")


class Comms(C.Structure):
    _fields_ = [("nsend", C.c_int), ("nrecv", C.c_int)] + \
               [(n, C.c_int * MAXCOMM) for n in _COMM_ARRAYS]

    def sends(self):
        return [dict(dir=self.dirsend[k], dest=self.destination[k],
                     isrc=self.isrcsend[k], jsrc=self.jsrcsend[k],
                     ides=self.idessend[k], jdes=self.jdessend[k],
                     nx=self.nxsend[k], ny=self.nysend[k]) for k in range(self.nsend)]

    def recvs(self):
        return [dict(dir=self.dirrecv[k], src=self.source[k],
                     isrc=self.isrcrecv[k], jsrc=self.jsrcrecv[k],
                     ides=self.idesrecv[k], jdes=self.jdesrecv[k],
                     nx=self.nxrecv[k], ny=self.nyrecv[k]) for k in range(self.nrecv)]


class SwParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fsdx", "fsdy", "tdts8", "tdtsdx", "tdtsdy")]


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    pI = C.POINTER(C.c_int)
    L.orc_grid_extents.argtypes = [C.c_int, C.c_int, C.c_int, pI, pI]
    L.orc_field_bounds.argtypes = [C.c_int] * 4 + [C.POINTER(Region), C.c_int, C.c_int,
                                                   C.POINTER(Region), C.POINTER(Region)]
    L.orc_field_bounds.restype = C.c_int
    L.orc_decompose.argtypes = [C.c_int] * 6 + [C.POINTER(Decomp), C.POINTER(Subdomain)]
    L.orc_iprocmap.argtypes = [C.POINTER(Decomp), C.POINTER(Subdomain), C.c_int, C.c_int, C.c_int]
    L.orc_iprocmap.restype = C.c_int
    L.orc_map_comms.argtypes = [C.POINTER(Decomp), C.POINTER(Subdomain), C.c_int, C.c_int,
                                C.POINTER(Comms)]
    L.orc_map_comms.restype = C.c_int
    L.orc_exchange_all.argtypes = [C.c_int, C.POINTER(C.c_void_p), pI, C.POINTER(Comms)]
    L.orc_exchange_all.restype = C.c_int
    L.orc_exchange_dirs.argtypes = [C.c_int, C.POINTER(C.c_void_p), pI, C.POINTER(Comms)] + [C.c_int] * 5
    L.orc_exchange_dirs.restype = C.c_int
    L.orc_checksum.argtypes = [_dp] + [C.c_int] * 5
    L.orc_checksum.restype = C.c_double
    L.orc_scatter.argtypes = [_dp, C.c_int, C.POINTER(Subdomain), _dp, C.c_int]
    L.orc_gather_all.argtypes = [C.c_int, C.POINTER(C.c_void_p), pI, C.POINTER(Decomp),
                                 C.POINTER(Subdomain), _dp]
    L.orc_hash_u01.argtypes = [C.c_uint64, C.c_int64, C.c_int64]
    L.orc_hash_u01.restype = C.c_double
    L.orc_jacobi5.argtypes = [_dp, _dp] + [C.c_int] * 5
    L.orc_jacobi5_omp.argtypes = [_dp, _dp] + [C.c_int] * 6
    _ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.orc_stencil9.argtypes = [_dp, _dp, _dp] + [C.c_int] * 5
    L.orc_continuity.argtypes = [C.c_double] + [C.c_int] * 5 + [_dp] * 9
    L.orc_continuity.restype = None
    L.orc_jacobi5_masked.argtypes = [_dp, _dp, _ip] + [C.c_int] * 5
    L.orc_tmask_fill.argtypes = [C.c_void_p] + [C.c_int] * 7 + [_ip]
    L.orc_sw_step.argtypes = [C.POINTER(SwParams)] + [C.c_int] * 5 + [_dp] * 13
    L.orc_sw_step_sw.argtypes = [C.POINTER(SwParams)] + [C.c_int] * 5 + [_dp] * 13
    L.orc_sw_kernel.argtypes = [C.c_int] * 7 + [C.c_double] * 2 + [C.c_void_p] * 5
    L.orc_sw_kernel.restype = C.c_int
    L.orc_periodic_halos.argtypes = [C.POINTER(Region), C.c_int, C.c_int, C.POINTER(Region), C.POINTER(Region)]
    L.orc_periodic_halos.restype = C.c_int
    L.orc_apply_periodic_halos.argtypes = [_dp, C.c_int, C.POINTER(Region), C.c_int, C.c_int]
    L.orc_max_threads.restype = C.c_int
    L.orc_copy_rows_omp.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


# ---- pythonic helpers -------------------------------------------------------
def grid_extents(sub_gnx, sub_gny, alignment=None):
    nx, ny = C.c_int(), C.c_int()
    lib().orc_grid_extents(sub_gnx, sub_gny, alignment or 0, C.byref(nx), C.byref(ny))
    return nx.value, ny.value


def decompose(domainx, domainy, ndom, ntilex=0, ntiley=0, hwidth=1):
    d = Decomp()
    subs = (Subdomain * ndom)()
    lib().orc_decompose(domainx, domainy, ndom, ntilex, ntiley, hwidth, C.byref(d), subs)
    return d, subs


def field_bounds(ptype, offset, bcx, bcy, sub_internal, grid_nx, grid_ny):
    i, w = Region(), Region()
    rc = lib().orc_field_bounds(ptype, offset, bcx, bcy, C.byref(sub_internal), grid_nx, grid_ny,
                                C.byref(i), C.byref(w))
    return rc, i, w


def map_comms(d, subs, nranks, irank):
    c = Comms()
    rc = lib().orc_map_comms(C.byref(d), subs, nranks, irank, C.byref(c))
    assert rc == 0, rc
    return c


def _ptr_array(fields):
    arr = (C.c_void_p * len(fields))()
    for k, f in enumerate(fields):
        assert f.dtype == np.float64 and f.flags["C_CONTIGUOUS"]
        arr[k] = f.ctypes.data
    return arr


def exchange_all(fields, lds, comms):
    """fields: list of C-contiguous (ny, ld) float64 arrays (row = j, the Fortran 2nd dim)."""
    n = len(fields)
    ldarr = (C.c_int * n)(*lds)
    carr = (Comms * n)(*comms)
    return lib().orc_exchange_all(n, _ptr_array(fields), ldarr, carr)


def exchange_dirs(fields, lds, comms, dirs, no_diagonals=False):
    """exchange_generic(comm1..comm4 = dirs) for every rank at once; diagonals follow their edges
    (parallel_comms_mod.f90:1557-1571) unless no_diagonals"""
    n = len(fields)
    c = (list(dirs) + [0, 0, 0, 0])[:4]
    return lib().orc_exchange_dirs(n, _ptr_array(fields), (C.c_int * n)(*lds), (Comms * n)(*comms),
                                   c[0], c[1], c[2], c[3], 1 if no_diagonals else 0)


def gather_all(fields, lds, d, subs):
    n = len(fields)
    out = np.zeros((d.global_ny, d.global_nx))
    lib().orc_gather_all(n, _ptr_array(fields), (C.c_int * n)(*lds), C.byref(d), subs, out)
    return out


def hash_field(seed, ny, ld, gx0, gy0, xlo, xhi, ylo, yhi):
    """(ny, ld) array, zero everywhere except local 1-based [xlo..xhi]x[ylo..yhi], which gets
    u01(splitmix64(seed ^ (gi + gj<<32))) with gi = gx0 + (i - 1), gj = gy0 + (j - 1).
    Vectorised numpy twin of orc_hash_u01 (checked against it in tests)."""
    f = np.zeros((ny, ld))
    i = np.arange(xlo, xhi + 1, dtype=np.uint64)
    j = np.arange(ylo, yhi + 1, dtype=np.uint64)
    gi = (np.uint64(gx0) + i - np.uint64(1))[None, :]
    gj = (np.uint64(gy0) + j - np.uint64(1))[:, None]
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (gi + (gj << np.uint64(32)))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    f[ylo - 1:yhi, xlo - 1:xhi] = (x >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return f


def host_threads():
    """threads this process may really use: cgroup quota / affinity / OpenMP limit (a GPU box
    gives a 1-GPU job a share of the host's cores, not all of them)"""
    n = min(lib().orc_max_threads(), len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_flib = None


def fortran_psy_lib():
    """oracle/libcpu_psy_fortran.so: the Jacobi step as Fortran kernel + PSy loop nest (amdflang)"""
    global _flib
    if _flib is None:
        so = os.path.join(ORACLE_DIR, "libcpu_psy_fortran.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libcpu_psy_fortran.so"], stdout=subprocess.DEVNULL)
        _flib = C.CDLL(so)
        _flib.psy_jacobi5_f.argtypes = [_dp, _dp] + [C.c_int] * 7
        _flib.psy_jacobi5_f.restype = None
        _flib.psy_shallow_step_f.argtypes = [_dp] + [C.c_int] * 6 + [_dp] * 13 + [C.c_int]
        _flib.psy_shallow_step_f.restype = None
    return _flib


def jacobi5_fortran(inp, out, ld, xs, xe, ys, ye, threads=1):
    """same update through the Fortran PSy loops (arrays are (ny, ld) C-order = (ld, ny) Fortran)"""
    fortran_psy_lib().psy_jacobi5_f(inp, out, ld, inp.shape[0], xs, xe, ys, ye, threads)


def jacobi5(inp, out, ld, xs, xe, ys, ye, threads=1):
    if threads > 1:
        lib().orc_jacobi5_omp(inp, out, ld, xs, xe, ys, ye, threads)
    else:
        lib().orc_jacobi5(inp, out, ld, xs, xe, ys, ye)


def sw_step(prm, ld, box, u, v, p, uold, vold, pold, unew, vnew, pnew):
    """orc_sw_step on the 1-based inclusive box (xs, xe, ys, ye); (ny, ld) arrays; the four
    intermediates live in scratch arrays of the same shape"""
    scratch = [np.zeros_like(p) for _ in range(4)]
    op = SwParams(prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy)
    xs, xe, ys, ye = box
    lib().orc_sw_step(C.byref(op), ld, xs, xe, ys, ye, u, v, p, uold, vold, pold, *scratch, unew, vnew, pnew)


def jacobi5_masked(inp, out, tmask, ld, xs, xe, ys, ye):
    lib().orc_jacobi5_masked(inp, out, tmask, ld, xs, xe, ys, ye)


def tmask_fill(user, nx, ny, internal):
    """grid_init's tmask (grid_mod.f90:394-455); user: (rows, cols) int32 array or None; internal =
    (xstart, xstop, ystart, ystop) of the subdomain"""
    out = np.zeros((ny, nx), dtype=np.int32)
    if user is None:
        lib().orc_tmask_fill(None, 0, nx, ny, *internal, out)
    else:
        u = np.ascontiguousarray(user, dtype=np.int32)
        lib().orc_tmask_fill(u.ctypes.data, u.shape[1], nx, ny, *internal, out)
    return out


def sw_step_sw(prm, ld, box, u, v, p, uold, vold, pold, unew, vnew, pnew):
    """orc_sw_step_sw (SW-offset staggering) on the 1-based inclusive box"""
    scratch = [np.zeros_like(p) for _ in range(4)]
    op = SwParams(prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy)
    xs, xe, ys, ye = box
    lib().orc_sw_step_sw(C.byref(op), ld, xs, xe, ys, ye, u, v, p, uold, vold, pold, *scratch, unew, vnew, pnew)


def periodic_halos(internal, bcx, bcy):
    """[(source as6, dest as6)] of init_periodic_bc_halos; internal = (xstart, xstop, ystart, ystop)"""
    it = Region(0, 0, *internal)
    src, dst = (Region * 4)(), (Region * 4)()
    n = lib().orc_periodic_halos(C.byref(it), bcx, bcy, src, dst)
    return [(src[k].as6()[:4], dst[k].as6()[:4]) for k in range(n)]


def apply_periodic_halos(f, ld, internal, bcx, bcy):
    it = Region(0, 0, *internal)
    lib().orc_apply_periodic_halos(f, ld, C.byref(it), bcx, bcy)


def continuity(rdt, ld, box, sshn_t, sshn_u, sshn_v, hu, hv, un, vn, area_t, ssha):
    lib().orc_continuity(rdt, ld, *box, sshn_t, sshn_u, sshn_v, hu, hv, un, vn, area_t, ssha)


def stencil9(inp, out, coef, ld, xs, xe, ys, ye):
    c = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(9))
    lib().orc_stencil9(inp, out, c, ld, xs, xe, ys, ye)


SW_KERNELS = ("cu", "cv", "z", "h", "unew", "vnew", "pnew", "time_smooth")


def sw_kernel(name, sw_offset, ld, box, out, ins, s0=0.0, s1=0.0):
    """ONE kernel of the GOcean shallow set as its own PSy loop nest over the 1-based inclusive box:
    out <- kern(ins...), arrays in the kernel's argument order (see orc_sw_kernel); (ny, ld) arrays"""
    ptr = [None] * 4
    for k, a in enumerate(ins):
        assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
        ptr[k] = a.ctypes.data
    assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
    rc = lib().orc_sw_kernel(SW_KERNELS.index(name), 1 if sw_offset else 0, ld, *box, s0, s1, out.ctypes.data, *ptr)
    assert rc == 0, rc


def sw_step_fortran(prm, ld, box, u, v, p, uold, vold, pold, unew, vnew, pnew, threads=1, scratch=None):
    """the NE-offset step as the seven Fortran PSy loop nests over pointwise GOcean kernels (oracle/cpu_psy_loops.f90),
    OpenMP over jj: what a GOcean application runs on the CPU"""
    sc = scratch if scratch is not None else [np.zeros_like(p) for _ in range(4)]
    pr = np.array([prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy])
    fortran_psy_lib().psy_shallow_step_f(pr, ld, p.shape[0], *box, u, v, p, uold, vold, pold, *sc, unew, vnew, pnew, threads)
