"""Python mirror of parallel_mod / parallel_utils_mod (reference:
finite_difference/src/parallel_mod.f90, parallel/parallel_utils_mod.f90).

Message passing is RCCL inside libdlesm_hip.so; this module only holds the rank
bookkeeping (1-based rank, as parallel_utils_mod.f90:84) and the bootstrap of the
RCCL communicator from a torch.distributed process group (used once, to hand the
ncclUniqueId around -- no data-path traffic goes through torch).
"""
import ctypes as C
import os

from . import _cabi
from ._cabi import CommTables, Decomp, Subdomain, check

_rank = 1      # get_rank(): 1-based
_nranks = 1
_rccl_up = False


class decomposition_type:
    """decomposition_mod.f90:54-68"""

    def __init__(self, info, subdomains):
        self._info = info
        self.global_nx, self.global_ny = info.global_nx, info.global_ny
        self.nx, self.ny = info.nx, info.ny
        self.ndomains = info.ndomains
        self.max_width, self.max_height = info.max_width, info.max_height
        self.subdomains = subdomains           # ctypes array of Subdomain, 0-based here
        # proc_subdomains(:, n): one subdomain per rank (parallel_mod.f90:141-152)
        self.proc_subdomains = [[r + 1] for r in range(info.ndomains)]


def parallel_init(rank=None, nranks=None, use_rccl=None, transport=None):
    """parallel_init (parallel_mod.f90:51-63).  rank is 0-based here (RANK env style) and
    stored 1-based.  With more than one rank the RCCL communicator is created; the unique id
    travels through the already-initialised torch.distributed group.
    transport="mailbox" (or DLESM_TRANSPORT=mailbox): no RCCL communicator -- every plan connects its mailboxes when it is
    made, exchanges and distributed steps are stores into the neighbours' memory (dlesm_comm_init_mailbox)."""
    global _rank, _nranks, _rccl_up
    if transport is None:
        transport = os.environ.get("DLESM_TRANSPORT", "rccl")
    if transport == "mailbox":
        if rank is None:
            rank = int(os.environ.get("RANK", "0"))
        if nranks is None:
            nranks = int(os.environ.get("WORLD_SIZE", "1"))
        _rank, _nranks = rank + 1, nranks
        if nranks > 1 and not _rccl_up:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("parallel_init: torch.distributed must be initialised to hand the session name round")
            L = _cabi.lib()
            buf = C.create_string_buffer(_cabi.UNIQUE_ID_BYTES)
            box = [None]
            if rank == 0:
                check(L.dlesm_board_nonce(buf))
                box[0] = buf.raw
            dist.broadcast_object_list(box, src=0)
            buf = C.create_string_buffer(box[0], _cabi.UNIQUE_ID_BYTES)
            check(L.dlesm_comm_init_mailbox(buf, nranks, rank))
            _rccl_up = True            # (a communicator of the other kind: parallel_finalise closes it)
        return
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    if nranks is None:
        nranks = int(os.environ.get("WORLD_SIZE", "1"))
    _rank, _nranks = rank + 1, nranks
    if use_rccl is None:
        use_rccl = nranks > 1
    if use_rccl and not _rccl_up:
        L = _cabi.lib()
        buf = C.create_string_buffer(_cabi.UNIQUE_ID_BYTES)
        if nranks > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("parallel_init: torch.distributed must be initialised to "
                                   "distribute the RCCL unique id")
            box = [None]
            if rank == 0:
                check(L.dlesm_comm_unique_id(buf))
                box[0] = buf.raw
            dist.broadcast_object_list(box, src=0)
            buf = C.create_string_buffer(box[0], _cabi.UNIQUE_ID_BYTES)
        else:
            check(L.dlesm_comm_unique_id(buf))
        check(L.dlesm_comm_init(buf, nranks, rank))
        _rccl_up = True


def parallel_finalise():
    global _rccl_up
    if _rccl_up:
        check(_cabi.lib().dlesm_comm_finalize())
        _rccl_up = False


def get_rank():
    return _rank


def get_num_ranks():
    return _nranks


def on_master():
    return _rank == 1


def go_decompose(domainx, domainy, ndomains=None, ndomainx=None, ndomainy=None, halo_width=1):
    """go_decompose (parallel_mod.f90:70-332)"""
    if ndomains is None:
        if ndomainx is None and ndomainy is None:
            ndomains = get_num_ranks()
        elif ndomainx is not None and ndomainy is not None:
            ndomains = ndomainx * ndomainy
        else:
            raise _cabi.GoceanStop(_cabi.EABORT, "go_decompose: invalid arguments supplied")
        tx, ty = (ndomainx or 0), (ndomainy or 0)
    else:
        tx = ty = 0                            # ndomains given => automatic tiling (pmod:123-126)
    info = Decomp()
    subs = (Subdomain * ndomains)()
    check(_cabi.lib().dlesm_decompose(domainx, domainy, ndomains, tx, ty, halo_width,
                                      C.byref(info), subs))
    return decomposition_type(info, subs)


def map_comms(decomp, rank1=None, nranks=None, depth=None):
    """map_comms (parallel_comms_mod.f90:178-1172) -> this rank's send/receive tables.
    depth=None: the reference's depth-1 tables; depth=d: the depth-d extension
    (dlesm_map_comms_depth; the reference aborts beyond MAX_HALO_DEPTH = 1)"""
    t = CommTables()
    L, n, r = _cabi.lib(), nranks or get_num_ranks(), rank1 or get_rank()
    if depth is None:
        check(L.dlesm_map_comms(C.byref(decomp._info), decomp.subdomains, n, r, C.byref(t)))
    else:
        check(L.dlesm_map_comms_depth(C.byref(decomp._info), decomp.subdomains, n, r, depth, C.byref(t)))
    return t
