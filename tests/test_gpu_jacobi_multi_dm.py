"""GPU test (-m gpu): the fused distributed Jacobi step (dlesm_jacobi5_multi_step_dm) with RCCL in
loop-back on one GPU -- rank 0 is its own eight neighbours, i.e. a periodic domain -- against
nsteps x (oracle step + oracle depth-nsteps exchange), bit for bit, halos included."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def D():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import dl_esm_inf_amd as d
    torch.cuda.set_device(0)
    d.parallel_init(0, 1, use_rccl=True)
    return d


CASES = [(4, 4, 2, 2), (2, 5, 2, 2), (9, 7, 3, 2), (40, 33, 2, 8), (257, 66, 4, 64), (130, 9, 4, None), (8, 8, 4, 2),
         (300, 40, 3, 64), (1900, 23, 4, 64), (64, 300, 2, None), (30, 20, 8, 2), (16, 16, 8, None), (700, 40, 6, 64),
         (64, 70, 5, 8), (33, 9, 7, 2), (1200, 64, 8, 64)]


@pytest.mark.parametrize("peer", [0, 1])
@pytest.mark.parametrize("nx,ny,nsteps,alignment", CASES)
def test_fused_distributed_step_in_loopback(D, nx, ny, nsteps, alignment, peer):
    """peer=1: the plan connected to the mailboxes -- the depth-nsteps exchanges (the initial one and the one inside the
    fused step) are the two-launch mailbox exchange, with the RCCL group switched off underneath"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny, halo_width=nsteps)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    it = a.internal
    assert it.xstart == nsteps + 1 and it.ystart == nsteps + 1 and it.nx == nx and it.ny == ny
    t = loopback_tables(D, it, nsteps)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    if peer:
        D.psy.halo_connect_peers(g)
        L.dlesm_set_tuning(b"dm_skip_parts", 1)        # no RCCL group: only the mailboxes can move the halos
    try:
        D.psy.hash_init(a, 77, box=D._cabi.Region(0, 0, 1, g.nx, 1, g.ny))
        before = a.get_data()
        a.halo_exchange(1)                     # depth comes from the plan's tables
        torch.cuda.synchronize()
        oc = O.Comms()
        C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
        cur = before.copy()
        assert O.exchange_all([cur], [g.nx], [oc]) == 0
        assert np.array_equal(a.get_data(), cur)
        D.copy_field(a, b)
        D.psy.invoke_jacobi5_multi_dm(b, a, nsteps)
        torch.cuda.synchronize()
        for _ in range(nsteps):
            nxt = cur.copy()
            O.jacobi5(cur, nxt, g.nx, *it.box())
            assert O.exchange_all([nxt], [g.nx], [oc]) == 0
            cur = nxt
        got = b.get_data()
        assert np.array_equal(got, cur), np.argwhere(got != cur)[:5]
        # a plan of another depth is refused
        if nsteps != 3 and nx >= 3 and ny >= 3:
            rc = L.dlesm_jacobi5_multi_step_dm(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, 3, *it.box(), None)
            assert rc == D._cabi.EINVAL
    finally:
        L.dlesm_set_tuning(b"dm_skip_parts", 0)
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))
        g._halo_plan = None
