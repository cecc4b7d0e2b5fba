"""ctypes binding of include/dlesm_hip.h (libdlesm_hip.so).

The library is built in-tree by `__graft_entry__.build()` / `make -C dl_esm_inf_amd/csrc`.
There is no fallback: if it is missing, importing this module raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DLESM_HIP_LIB: another build of the same library, for A/B measurements of kernel variants
LIB_PATH = os.environ.get("DLESM_HIP_LIB") or os.path.join(HERE, "lib", "libdlesm_hip.so")
MAXCOMM = 16
UNIQUE_ID_BYTES = 128

OK, EINVAL, ENODEV, EHIP, ERCCL, EABORT, ECOMMS = 0, -1, -2, -3, -4, -5, -12
# dirs_mask of dlesm_halo_exchange_f64: bit d-1 per edge direction; 0 exchanges nothing
PEER_BLOB_BYTES = 1024
DIRS_ALL, DIRS_NO_DIAGONALS = 0xF, 0x10
DIRS_EDGES_ONLY = DIRS_ALL | DIRS_NO_DIAGONALS


class DlesmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dlesm error {code}: {msg}")
        self.code = code


class GoceanStop(DlesmError):
    """the reference calls gocean_stop() (fatal) for this input"""


class Region(C.Structure):
    """region_mod.f90:7-12"""
    _fields_ = [(n, C.c_int) for n in ("nx", "ny", "xstart", "xstop", "ystart", "ystop")]

    def as6(self):
        return [self.xstart, self.xstop, self.ystart, self.ystop, self.nx, self.ny]

    def box(self):
        return (self.xstart, self.xstop, self.ystart, self.ystop)

    def __repr__(self):
        return f"Region(x={self.xstart}:{self.xstop}, y={self.ystart}:{self.ystop}, n={self.nx}x{self.ny})"


class Subdomain(C.Structure):
    """decomposition_mod.f90:44-50 (`global` is a Python keyword, hence glob)"""
    _fields_ = [("glob", Region), ("internal", Region)]


class Decomp(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("global_nx", "global_ny", "nx", "ny", "ndomains",
                                       "max_width", "max_height")]


_COMM_ARRAYS = ("dirsend", "destination", "isrcsend", "jsrcsend", "idessend", "jdessend",
                "nxsend", "nysend", "dirrecv", "source", "isrcrecv", "jsrcrecv",
                "idesrecv", "jdesrecv", "nxrecv", "nyrecv")


class CommTables(C.Structure):
    """public tables of parallel_comms_mod.f90:71-83"""
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


class MsgDesc(C.Structure):
    """dlesm_msg_desc: one ncclRecv / ncclSend of an exchange, in issue order"""
    _fields_ = [(n, C.c_int) for n in ("is_recv", "peer", "dir", "field", "i0", "j0", "nx", "ny")] + \
               [("count", C.c_long), ("buffer_offset", C.c_long)]


class SwParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fsdx", "fsdy", "tdts8", "tdtsdx", "tdtsdy")]


# every entry point include/dlesm_hip.h declares: name -> (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_pi = C.POINTER(C.c_int)
PROTOTYPES = {
    "dlesm_alignment_from_env": (_i, [_pi]),
    "dlesm_grid_extents": (_i, [_i, _i, _i, _pi, _pi]),
    "dlesm_field_bounds": (_i, [_i, _i, _i, _i, C.POINTER(Region), _i, _i,
                                C.POINTER(Region), C.POINTER(Region)]),
    "dlesm_decompose": (_i, [_i] * 6 + [C.POINTER(Decomp), C.POINTER(Subdomain)]),
    "dlesm_iprocmap": (_i, [C.POINTER(Decomp), C.POINTER(Subdomain), _i, _i, _i]),
    "dlesm_map_comms": (_i, [C.POINTER(Decomp), C.POINTER(Subdomain), _i, _i,
                             C.POINTER(CommTables)]),
    "dlesm_map_comms_depth": (_i, [C.POINTER(Decomp), C.POINTER(Subdomain), _i, _i, _i,
                                   C.POINTER(CommTables)]),
    "dlesm_last_error": (C.c_char_p, []),
    "dlesm_version": (_i, []),
    "dlesm_device_count": (_i, []),
    "dlesm_init": (_i, [_i]),
    "dlesm_finalize": (_i, []),
    "dlesm_field_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "dlesm_field_wrap": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "dlesm_field_destroy": (_i, [_vp]),
    "dlesm_field_data": (_vp, [_vp]),
    "dlesm_field_ld": (_i, [_vp]),
    "dlesm_field_ny": (_i, [_vp]),
    "dlesm_read_from_device": (None, [_vp, _vp, _i, _i, _i, _i, C.c_bool]),
    "dlesm_write_to_device": (None, [_vp, _vp, _i, _i, _i, _i, C.c_bool]),
    "dlesm_transfer_sync": (_i, []),
    "dlesm_stencil5_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_stencil5_planned_shape": (_i, [_i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "dlesm_continuity_f64": (_i, [_d, _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_stencil9_f64": (_i, [_vp, _vp, C.POINTER(_d), _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_stencil9_step_dm": (_i, [_vp, _vp, _vp, C.POINTER(_d), _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_stencil5_masked_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_stencil5_autotune_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_stencil5_x2_f64": (_i, [_vp, _vp] + [_i] * 10 + [_vp]),
    "dlesm_stencil5_multi_f64": (_i, [_vp, _vp] + [_i] * 15 + [_vp]),
    "dlesm_shallow_step_f64": (_i, [C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_autotune_f64": (_i, [C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_autotune_sw_f64": (_i, [C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_smooth_f64": (_i, [C.POINTER(SwParams), _d, _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_sw_smooth_periodic_f64": (_i, [C.POINTER(SwParams), _d, _i, _i, C.POINTER(Region), _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_compute_cu_f64": (_i, [_i] * 7 + [_vp] * 3 + [_vp]),
    "dlesm_compute_cv_f64": (_i, [_i] * 7 + [_vp] * 3 + [_vp]),
    "dlesm_compute_z_f64": (_i, [_i] * 7 + [_d, _d] + [_vp] * 4 + [_vp]),
    "dlesm_compute_h_f64": (_i, [_i] * 7 + [_vp] * 4 + [_vp]),
    "dlesm_compute_unew_f64": (_i, [_i] * 7 + [_d, _d] + [_vp] * 5 + [_vp]),
    "dlesm_compute_vnew_f64": (_i, [_i] * 7 + [_d, _d] + [_vp] * 5 + [_vp]),
    "dlesm_compute_pnew_f64": (_i, [_i] * 7 + [_d, _d] + [_vp] * 4 + [_vp]),
    "dlesm_time_smooth_f64": (_i, [_i] * 6 + [_d] + [_vp] * 3 + [_vp]),
    "dlesm_periodic_halos": (_i, [C.POINTER(Region), _i, _i, C.POINTER(Region), C.POINTER(Region), C.POINTER(_i)]),
    "dlesm_periodic_halos_apply_f64": (_i, [_vp, _i, _i, C.POINTER(Region), _i, _i, _vp]),
    "dlesm_periodic_halos_apply_multi_f64": (_i, [C.POINTER(_vp), _i, _i, _i, C.POINTER(Region), _i, _i, _vp]),
    "dlesm_shallow_step_sw_f64": (_i, [C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_sw_periodic_f64": (_i, [C.POINTER(SwParams), _i, _i, C.POINTER(Region), _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_copy_patch_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_fill_f64": (_i, [_vp, _i, _i, _i, _i, _i, _i, _d, _vp]),
    "dlesm_checksum_f64": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_d), _vp]),
    "dlesm_checksum_async_f64": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "dlesm_hash_init_f64": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.c_uint64, C.c_int64, C.c_int64, _vp]),
    "dlesm_set_tuning": (_i, [C.c_char_p, _i]),
    "dlesm_tuning_class": (_i, [C.c_char_p]),
    "dlesm_is_lab_build": (_i, []),
    "dlesm_comm_unique_id": (_i, [_vp]),
    "dlesm_comm_init": (_i, [_vp, _i, _i]),
    "dlesm_comm_finalize": (_i, []),
    "dlesm_rendezvous_remove": (_i, [C.c_char_p]),
    "dlesm_rendezvous_publish": (_i, [C.c_char_p, _vp, C.c_char_p]),
    "dlesm_rendezvous_fetch": (_i, [C.c_char_p, _vp, C.c_char_p, _i]),
    "dlesm_rendezvous_ack": (_i, [C.c_char_p, _i]),
    "dlesm_rendezvous_wait_acks": (_i, [C.c_char_p, _i, _i]),
    "dlesm_comm_rank": (_i, []),
    "dlesm_comm_size": (_i, []),
    "dlesm_halo_plan_create": (_i, [C.POINTER(CommTables), _i, _i, C.POINTER(_vp)]),
    "dlesm_halo_plan_destroy": (_i, [_vp]),
    "dlesm_halo_plan_describe": (_i, [C.POINTER(CommTables), _i, _i, _i, C.c_uint, _i, C.POINTER(MsgDesc), _i, C.POINTER(_i)]),
    "dlesm_halo_exchange_f64": (_i, [_vp, _vp, C.c_uint, _vp]),
    "dlesm_halo_exchange_multi_f64": (_i, [_vp, C.POINTER(_vp), _i, C.c_uint, _vp]),
    "dlesm_jacobi5_step_dm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_jacobi5_step_dm_pipelined": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_halo_plan_join": (_i, [_vp, _vp]),
    "dlesm_comm_init_mailbox": (_i, [_vp, _i, _i]),
    "dlesm_comm_is_mailbox": (_i, []),
    "dlesm_ipc_open_retries": (_i, []),
    "dlesm_board_nonce": (_i, [_vp]),
    "dlesm_board_open": (_i, [_vp, _i, _i]),
    "dlesm_board_is_open": (_i, []),
    "dlesm_board_allgather": (_i, [_vp, C.c_size_t, _vp]),
    "dlesm_board_close": (_i, []),
    "dlesm_board_abort": (_i, [C.c_char_p]),
    "dlesm_halo_plan_peer_export": (_i, [_vp, _i, _i, _vp]),
    "dlesm_halo_plan_peer_connect": (_i, [_vp, _i, _i, _vp]),
    "dlesm_halo_plan_peer_connect_rccl": (_i, [_vp, _i]),
    "dlesm_halo_plan_peer_connected": (_i, [_vp]),
    "dlesm_peer_blob_describe": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "dlesm_peer_match_describe": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, C.POINTER(_i)]),
    "dlesm_wait_timed_out": (_i, [_i]),
    "dlesm_probe_stream_concurrency": (_i, [_vp]),
    "dlesm_jacobi5_multi_step_dm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dlesm_shallow_step_x2_f64": (_i, [C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 12 + [_vp]),
    "dlesm_shallow_step_smooth_x2_f64": (_i, [C.POINTER(SwParams), _d, _i, _i, _i, _i, _i, _i] + [_vp] * 12 + [_vp]),
    "dlesm_shallow_step_sw_x2_periodic_f64": (_i, [C.POINTER(SwParams), _i, _i, C.POINTER(Region), _i, _i] + [_vp] * 12 + [_vp]),
    "dlesm_shallow_step_sw_smooth_x2_periodic_f64": (_i, [C.POINTER(SwParams), _d, _i, _i, C.POINTER(Region), _i, _i] + [_vp] * 12 + [_vp]),
    "dlesm_shallow_step_dm": (_i, [_vp, C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_dm_pipelined": (_i, [_vp, C.POINTER(SwParams), _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_smooth_dm": (_i, [_vp, C.POINTER(SwParams), _d, _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_shallow_step_smooth_dm_pipelined": (_i, [_vp, C.POINTER(SwParams), _d, _i, _i, _i, _i, _i, _i] + [_vp] * 9 + [_vp]),
    "dlesm_global_sum_f64": (_i, [C.POINTER(_d)]),
    "dlesm_gather_f64": (_i, [_vp, _vp, _i]),
    "dlesm_pack_inner_f64": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, C.c_long, _vp]),
    "dlesm_unpack_gathered_f64": (_i, [_vp, C.c_long, C.POINTER(Decomp), C.POINTER(Subdomain), _i, _vp, _vp]),
    "dlesm_gather_inner_f64": (_i, [_vp, _i, _i, C.POINTER(Region), C.POINTER(Decomp), C.POINTER(Subdomain), _i, _vp]),
    "dlesm_scatter_inner_f64": (_i, [_vp, _i, _i, C.POINTER(Subdomain), _vp, _i, _i]),
}

_lib = None


def lib():
    """load libdlesm_hip.so (after torch, so that the HIP/RCCL runtimes are shared)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C dl_esm_inf_amd/csrc`). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64.so.7 / librccl.so.1 first)
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)         # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ---- measurement tooling (include/dlesm_lab.h): loaded NEXT TO the product library by bench.py, scripts/ and tests/ ----
LAB_LIB_PATH = os.path.join(HERE, "lib", "libdlesm_lab.so")
# the measurement build of the product's own sources (-DDLESM_LAB): what DLESM_HIP_LIB is pointed at to reach the
# comparison-only kernels (LAB_NOTES.md); never loaded unless asked for
LAB_BUILD_PATH = os.path.join(HERE, "lib", "libdlesm_hip_lab.so")
LAB_PROTOTYPES = {
    "dlesm_lab_last_error": (C.c_char_p, []),
    "dlesm_lab_stream_copy_f64": (_i, [_i, _i, C.POINTER(_vp), C.POINTER(_vp), C.c_size_t, _i, _vp]),
}
_lab = None


def lab():
    """load libdlesm_lab.so: the copy-ceiling sweeps (no state shared with the product library)"""
    global _lab
    if _lab is not None:
        return _lab
    if not os.path.exists(LAB_LIB_PATH):
        raise ImportError(f"{LAB_LIB_PATH} is missing (make -C dl_esm_inf_amd/csrc lab)")
    lib()                                # the HIP runtime is in the process
    L = C.CDLL(LAB_LIB_PATH)
    for name, (res, args) in LAB_PROTOTYPES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lab = L
    return L


def check_lab(rc):
    if rc != 0:
        raise DlesmError(rc, lab().dlesm_lab_last_error().decode(errors="replace"))


def is_lab_build():
    """True when the library in this process is libdlesm_hip_lab.so (DLESM_HIP_LIB pointed at it)"""
    return bool(lib().dlesm_is_lab_build())


class PeerMatchDesc(C.Structure):
    _fields_ = [("peer", _i), ("dir", _i), ("i0", _i), ("j0", _i), ("nx", _i), ("ny", _i), ("count", C.c_long), ("slot", _i),
                ("off", C.c_long)]


def last_error():
    """the message of the last failed call on this thread"""
    return lib().dlesm_last_error().decode(errors="replace")


def check(rc):
    if rc == OK:
        return
    msg = lib().dlesm_last_error().decode(errors="replace")
    if rc == EABORT:
        raise GoceanStop(rc, msg)
    raise DlesmError(rc, msg)
