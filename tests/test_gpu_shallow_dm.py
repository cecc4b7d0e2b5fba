"""GPU test (-m gpu): the distributed shallow-water step and the multi-field grouped exchange, with
RCCL in loop-back on one GPU (rank 0 is its own eight neighbours), against the oracle's step
followed by the oracle's exchange of the three new fields."""
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


@pytest.mark.parametrize("nx,ny,alignment", [(1, 1, 2), (2, 3, 2), (5, 1, 2), (1, 7, 2), (40, 33, 8), (257, 66, 64),
                                             (130, 9, None), (700, 300, 64)])
@pytest.mark.parametrize("one_launch_frame,fused,aggregate,peer", [(1, 1, 1, 0), (1, 0, 1, 0), (0, 0, 1, 0), (1, 1, 0, 0),
                                                                   (0, 0, 0, 0), (1, 1, 1, 1), (1, 0, 1, 1), (1, 1, 1, 2)])
def test_shallow_step_dm_matches_oracle(D, nx, ny, alignment, one_launch_frame, fused, aggregate, peer):
    """peer=1: the plan connected to the mailboxes for three fields -- the ring workgroups store into the
    neighbour's mailbox, no RCCL kernel in the step (fused=0: the ring in its own launch, its flags behind it).  fused: the ring as the first workgroups of the interior launch + device flag (default) /
    one_launch_frame: the ring in its own launch that also fills the send buffer / the round-1 form,
    four thin boxes + pack kernels.  aggregate=0: one message per field and direction (24 instead of 8),
    the form before the aggregated exchange, kept as the comparison point."""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    L.dlesm_set_tuning(b"sw_dm_frame", one_launch_frame)
    L.dlesm_set_tuning(b"sw_dm_fused", fused)
    L.dlesm_set_tuning(b"dm_aggregate", aggregate)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 0 if peer == 2 else 1)    # peer=2: the join as its own wait + unpack launch
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
    it = F["p"].internal
    # the loop-back plan replaces the (empty) serial tables of this grid
    t = loopback_tables(D, it)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    if peer:
        D.psy.halo_connect_peers(g, 3)
    for k, n in enumerate(names[:6]):
        D.psy.hash_init(F[n], 100 + k)
        F[n].data.add_(1.0 if n[0] == "p" else -0.5)
    for n in names[6:]:
        D.set_field(F[n], 9.0)
    D.psy.halo_exchange_multi([F["u"], F["v"], F["p"]])        # inputs with wrapped halos
    torch.cuda.synchronize()
    H = {n: F[n].get_data() for n in names}
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)

    # oracle: the multi-field exchange just done, then step + exchange of the new fields
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    ref_in = {}
    for k, n in enumerate(names[:3]):
        f = O.hash_field(100 + k, g.ny, g.nx, 0, 0, 1, nx + 2, 1, ny + 2)
        f += 1.0 if n == "p" else -0.5
        assert O.exchange_all([f], [g.nx], [oc]) == 0
        ref_in[n] = f
        assert np.array_equal(H[n], f), n
    want = {n: np.full((g.ny, g.nx), 9.0) for n in names[6:]}
    scratch = [np.zeros((g.ny, g.nx)) for _ in range(4)]
    op = O.SwParams(prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy)
    O.lib().orc_sw_step(C.byref(op), g.nx, *it.box(), ref_in["u"], ref_in["v"], ref_in["p"],
                        H["uold"], H["vold"], H["pold"], *scratch, want["unew"], want["vnew"], want["pnew"])
    for n in names[6:]:
        assert O.exchange_all([want[n]], [g.nx], [oc]) == 0

    D.psy.invoke_shallow_step_dm(prm, *[F[n] for n in names])
    torch.cuda.synchronize()
    for n in names[6:]:
        assert np.array_equal(F[n].get_data(), want[n]), n
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    g._halo_plan = None
    L.dlesm_set_tuning(b"sw_dm_frame", 1)
    L.dlesm_set_tuning(b"sw_dm_fused", 1)
    L.dlesm_set_tuning(b"dm_aggregate", 1)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 1)


@pytest.mark.parametrize("nx,ny,alignment,nsteps", [(40, 33, 8, 7), (257, 66, 64, 5), (130, 9, None, 6), (700, 300, 64, 9),
                                                    (2, 3, 2, 4)])
@pytest.mark.parametrize("chain,peer", [(1, 0), (0, 0), (1, 1), (1, 2)])
def test_pipelined_shallow_time_loop(D, nx, ny, alignment, nsteps, chain, peer):
    """peer=1: the same loop over the mailboxes (plan connected for three fields).  a leapfrog time loop of dlesm_shallow_step_dm_pipelined (three time levels rotated by pointer,
    the exchange of step k joined on the device by step k+1's frame workgroups, one join at the end)
    against the oracle's step + exchange, every field and halo, bit for bit.  chain=0: the same calls
    with the device-side join switched off (event join at the start of each step)."""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    L.dlesm_set_tuning(b"sw_dm_chain", chain)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 0 if peer == 2 else 1)    # peer=2: separate wait + unpack launch behind each step
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
    it = F["p"].internal
    t = loopback_tables(D, it)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    if peer:
        D.psy.halo_connect_peers(g, 3)
    for k, n in enumerate(names[:6]):
        D.psy.hash_init(F[n], 200 + k)
        F[n].data.mul_(0.01)
        F[n].data.add_(1.0 if n[0] == "p" else -0.005)
    for n in names[6:]:
        D.set_field(F[n], 9.0)
    D.psy.halo_exchange_multi([F[n] for n in names[:6]])
    torch.cuda.synchronize()
    H = {n: F[n].get_data() for n in names}
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    op = O.SwParams(prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy)
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    scratch = [np.zeros((g.ny, g.nx)) for _ in range(4)]

    cur, old, new = ["u", "v", "p"], ["uold", "vold", "pold"], ["unew", "vnew", "pnew"]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(nsteps):
            D.psy.invoke_shallow_step_dm_pipelined(prm, *[F[n] for n in cur + old + new], stream=s)
            O.lib().orc_sw_step(C.byref(op), g.nx, *it.box(), *[H[n] for n in cur + old], *scratch, *[H[n] for n in new])
            for n in new:
                assert O.exchange_all([H[n]], [g.nx], [oc]) == 0
            cur, old, new = new, cur, old                # leapfrog rotation: pointers only
        D.psy.halo_join(g, stream=s)
    s.synchronize()
    for n in names:
        assert np.array_equal(F[n].get_data(), H[n]), n
    assert np.isfinite(H[cur[2]]).all()
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    g._halo_plan = None
    L.dlesm_set_tuning(b"sw_dm_chain", 1)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 1)


def test_shallow_dm_at_the_weak_scaling_tile(D):
    """8192^2 (the per-GPU tile of BASELINE configs[4]) in loop-back: three leapfrog steps in the time-loop form
    + one join equal three times (plain step, then grouped exchange) on the same device, every field and halo,
    bit for bit; two sampled rows of the first step against an oracle slab"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    n = 8192
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(n, n)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    A = {k: D.r2d_field(g, pts[k[0]]) for k in names}       # time-loop form
    B = {k: D.r2d_field(g, pts[k[0]]) for k in names}       # step, then exchange
    it = A["p"].internal
    t = loopback_tables(D, it)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    for k, name in enumerate(names):
        D.psy.hash_init(A[name], 300 + k)
        A[name].data.mul_(0.01)
        A[name].data.add_(1.0 if name[0] == "p" else -0.005)
    D.psy.halo_exchange_multi([A[k] for k in names[:6]])
    for name in names:
        D.copy_field(A[name], B[name])
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    # oracle slab of the first step: rows j0 .. j0+1 of the new fields
    j0 = it.ystart + 4000
    slab = [A[k].data[j0 - 2:j0 + 2, :].cpu().numpy() for k in names[:6]]
    want = [np.zeros_like(slab[0]) for _ in range(3)]
    O.sw_step(prm, g.nx, (it.xstart, it.xstop, 2, 3), *slab, *want)

    def order(F, c, o, nw):
        return [F[k] for k in c + o + nw]

    cur, old, new = names[:3], names[3:6], names[6:]
    for step in range(3):
        D.psy.invoke_shallow_step_dm_pipelined(prm, *order(A, cur, old, new))
        D.psy.invoke_shallow_step(prm, *order(B, cur, old, new))
        D.psy.halo_exchange_multi([B[k] for k in new])
        if step == 0:
            D.psy.halo_join(g)
            torch.cuda.synchronize()
            for k, w in zip(new, want):
                assert np.array_equal(A[k].data[j0 - 1:j0 + 1, it.xstart - 1:it.xstop].cpu().numpy(),
                                      w[1:3, it.xstart - 1:it.xstop]), k
        cur, old, new = new, cur, old
    D.psy.halo_join(g)
    torch.cuda.synchronize()
    w = A["p"].whole
    for k in names:
        assert torch.equal(A[k].data[w.ystart - 1:w.ystop, w.xstart - 1:w.xstop],
                           B[k].data[w.ystart - 1:w.ystop, w.xstart - 1:w.xstop]), k
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    g._halo_plan = None


@pytest.mark.parametrize("peer", [0, 1])
def test_time_loop_forms_of_both_steps_share_a_plan(D, peer):
    """(peer=1: both kinds of step over the same mailboxes, one sequence number.)  a pipelined shallow-water step, then a pipelined Jacobi step on its pnew (chained on the device to the
    shallow step's exchange), then a pipelined shallow step again (which must first join the Jacobi step's
    un-unpacked exchange): one plan, one stream, every hand-over between the two kinds of step"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    nx, ny = 257, 66
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
    q = D.r2d_field(g, D.GO_T_POINTS)
    it = F["p"].internal
    t = loopback_tables(D, it)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    if peer:
        D.psy.halo_connect_peers(g, 3)
    for k, n in enumerate(names[:6]):
        D.psy.hash_init(F[n], 400 + k)
        F[n].data.mul_(0.01)
        F[n].data.add_(1.0 if n[0] == "p" else -0.005)
    for n in names[6:]:
        D.set_field(F[n], 9.0)
    D.set_field(q, 5.0)
    D.psy.halo_exchange_multi([F[n] for n in names[:6]])
    torch.cuda.synchronize()
    H = {n: F[n].get_data() for n in names}
    hq = q.get_data()
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))

    def shallow(cur, old, new):
        D.psy.invoke_shallow_step_dm_pipelined(prm, *[F[n] for n in cur + old + new])
        O.sw_step(prm, g.nx, it.box(), *[H[n] for n in cur + old], *[H[n] for n in new])
        for n in new:
            assert O.exchange_all([H[n]], [g.nx], [oc]) == 0

    cur, old, new = names[:3], names[3:6], names[6:]
    shallow(cur, old, new)
    # Jacobi on pnew: its frame workgroups wait on the device for the shallow step's exchange
    D.psy.invoke_jacobi5_dm_pipelined(q, F["pnew"])
    O.jacobi5(H["pnew"], hq, g.nx, *it.box())
    assert O.exchange_dirs([hq], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
    # the next shallow step joins the Jacobi step's exchange (and its deferred unpack into q) first
    shallow(new, cur, old)
    D.psy.halo_join(g)
    torch.cuda.synchronize()
    for n in names:
        assert np.array_equal(F[n].get_data(), H[n]), n
    assert np.array_equal(q.get_data(), hq)
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    g._halo_plan = None


@pytest.mark.parametrize("nx,ny,alignment", [(2, 3, 2), (40, 33, 8), (257, 66, 64), (130, 9, None), (700, 300, 64)])
@pytest.mark.parametrize("one_launch_frame,fused,pipelined,peer", [(1, 1, True, 0), (1, 1, False, 0), (1, 0, False, 0),
                                                                   (0, 0, False, 0), (1, 1, True, 1), (1, 0, False, 1), (1, 1, False, 2)])
def test_distributed_step_with_the_filter_folded_in(D, nx, ny, alignment, one_launch_frame, fused, pipelined, peer):
    """(peer=1: over the mailboxes.)  dlesm_shallow_step_smooth_dm[_pipelined] -- the distributed step that also filters the old level in place (Asselin,
    time_smooth) -- in a four-step time loop with the benchmark's rotation, RCCL in loop-back, against the oracle's step +
    exchange of the new level + time_smooth of the old level, every field and halo, bit for bit; in the one-launch,
    own-frame-launch and four-thin-boxes forms"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    L.dlesm_set_tuning(b"sw_dm_frame", one_launch_frame)
    L.dlesm_set_tuning(b"sw_dm_fused", fused)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 0 if peer == 2 else 1)    # peer=2: the join as its own wait + unpack launch
    try:
        if alignment is None:
            os.environ.pop("DL_ESM_ALIGNMENT", None)
        else:
            os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
        g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
        g.decompose(nx, ny)
        D.grid_init(g, 1.0e5, 1.0e5)
        os.environ.pop("DL_ESM_ALIGNMENT", None)
        names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
        pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
        F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
        it = F["p"].internal
        t = loopback_tables(D, it)
        plan = C.c_void_p()
        D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
        g._halo_plan = plan
        if peer:
            D.psy.halo_connect_peers(g, 3)
        for k, n in enumerate(names[:6]):
            D.psy.hash_init(F[n], 300 + k)
            F[n].data.mul_(0.01)
            F[n].data.add_(1.0 if n[0] == "p" else -0.005)
        for n in names[6:]:
            D.set_field(F[n], 9.0)
        D.psy.halo_exchange_multi([F[n] for n in names[:6]])
        torch.cuda.synchronize()
        H = {n: F[n].get_data() for n in names}
        alpha = 0.001
        prm = D.psy.shallow_params(g.dx, g.dy, 20.0)
        oc = O.Comms()
        C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
        cur, old, new = names[:3], names[3:6], names[6:]
        for step in range(4):
            D.psy.invoke_shallow_step_smooth_dm(prm, alpha, *[F[n] for n in cur + old + new], pipelined=pipelined)
            O.sw_step(prm, g.nx, it.box(), *[H[n] for n in cur + old + new])
            for n in new:
                assert O.exchange_all([H[n]], [g.nx], [oc]) == 0
            for c, nw, o in zip(cur, new, old):
                O.sw_kernel("time_smooth", False, g.nx, it.box(), H[o], [H[c], H[nw], H[o]], alpha)
            cur, new = new, cur
        D.psy.halo_join(g)
        torch.cuda.synchronize()
        for n in names:
            assert np.array_equal(F[n].get_data(), H[n]), n
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))
        g._halo_plan = None
    finally:
        L.dlesm_set_tuning(b"sw_dm_frame", 1)
        L.dlesm_set_tuning(b"sw_dm_fused", 1)
        L.dlesm_set_tuning(b"dm_peer_join_fused", 1)


@pytest.mark.parametrize("nx,ny,alignment", [(257, 66, 64), (130, 9, None), (700, 300, 64)])
@pytest.mark.parametrize("filtered", [False, True])
def test_shallow_time_loop_over_the_mailboxes_captured_into_a_graph(D, nx, ny, alignment, filtered):
    """Six leapfrog steps of the distributed shallow-water step over the mailboxes (the three-level pointer rotation
    repeats after three steps, the mailbox halves after two) captured into ONE hipGraph and replayed three times:
    18 steps, against the oracle's step + exchange of the new level (+ time_smooth of the old level: the filtered form), every field and
    halo, bit for bit.  The steps' sequence numbers live on the device (peer_seq_load); no RCCL call is in the graph."""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {n: D.r2d_field(g, pts[n[0]]) for n in names}
    it = F["p"].internal
    t = loopback_tables(D, it)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    g._halo_plan = plan
    D.psy.halo_connect_peers(g, 3)
    s = torch.cuda.Stream()
    alpha = 0.001
    with torch.cuda.stream(s):
        for k, n in enumerate(names[:6]):
            D.psy.hash_init(F[n], 300 + k, stream=s)
            F[n].data.mul_(0.01)
            F[n].data.add_(1.0 if n[0] == "p" else -0.005)
        for n in names[6:]:
            D.set_field(F[n], 9.0, stream=s)
        D.psy.halo_exchange_multi([F[n] for n in names[:3]], stream=s)        # mailbox operations before the capture: on the capturing stream
        D.psy.halo_exchange_multi([F[n] for n in names[3:6]], stream=s)
    s.synchronize()
    H = {n: F[n].get_data() for n in names}
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    op = O.SwParams(prm.fsdx, prm.fsdy, prm.tdts8, prm.tdtsdx, prm.tdtsdy)
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    scratch = [np.zeros((g.ny, g.nx)) for _ in range(4)]
    graph = torch.cuda.CUDAGraph()
    start = (["u", "v", "p"], ["uold", "vold", "pold"], ["unew", "vnew", "pnew"])
    cur, old, new = start

    def rotate(cur, old, new):       # the filter leaves the filtered present level in `old`: two buffers swap; without it three rotate
        return (new, old, cur) if filtered else (new, cur, old)

    with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
        for _ in range(6):
            if filtered:
                D.psy.invoke_shallow_step_smooth_dm(prm, alpha, *[F[n] for n in cur + old + new], stream=s)
            else:
                D.psy.invoke_shallow_step_dm_pipelined(prm, *[F[n] for n in cur + old + new], stream=s)
            cur, old, new = rotate(cur, old, new)
        D.psy.halo_join(g, stream=s)
    assert (cur, old, new) == start
    for _ in range(3):
        with torch.cuda.stream(s):            # (replay() launches on the current stream)
            graph.replay()
        for _ in range(6):
            O.lib().orc_sw_step(C.byref(op), g.nx, *it.box(), *[H[n] for n in cur + old], *scratch, *[H[n] for n in new])
            for n in new:
                assert O.exchange_all([H[n]], [g.nx], [oc]) == 0
            if filtered:
                for c, nw, o in zip(cur, new, old):
                    O.sw_kernel("time_smooth", False, g.nx, it.box(), H[o], [H[c], H[nw], H[o]], alpha)
            cur, old, new = rotate(cur, old, new)
    torch.cuda.synchronize()
    assert L.dlesm_wait_timed_out(0) == 0
    for n in names:
        assert np.array_equal(F[n].get_data(), H[n]), n
    del graph
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    g._halo_plan = None
