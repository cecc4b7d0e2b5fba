"""GPU tests (-m gpu) of the PEER TRANSPORT of the distributed Jacobi step (dlesm_halo_plan_peer_*):
the frame workgroups of the step launch store into the neighbours' mailboxes and raise their arrival flags -- no RCCL
kernel.  One GPU: rank 0 is its own four (or eight) neighbours, so mailbox addressing, the matching of sends with
receives, the double buffering on the step number, the chained waits and the joins all run as they do between GPUs, minus
the IPC mapping (tests/peer_two_ranks_worker.py covers that with two processes) and the xGMI hop.

Oracle: steps x (orc_jacobi5 + the oracle's edge exchange on the same tables), every bit, halos included."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 20261004
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.fixture(scope="module")
def D():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    import dl_esm_inf_amd as d
    torch.cuda.set_device(0)
    d.parallel_init(0, 1, use_rccl=True)
    return d


def _grid(D, nx, ny, alignment):
    if alignment is None:
        os.environ.pop("DL_ESM_ALIGNMENT", None)
    else:
        os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    return g


def _setup(D, nx, ny, alignment, connect="rccl"):
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    g = _grid(D, nx, ny, alignment)
    x, y = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    t = loopback_tables(D, x.internal)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    assert L.dlesm_halo_plan_peer_connected(plan) == 0
    if connect == "rccl":          # export + ncclAllGather + connect inside the library
        D._cabi.check(L.dlesm_halo_plan_peer_connect_rccl(plan, 1))
    elif connect == "host":        # the three-step form a host program with its own all-gather uses
        blob = C.create_string_buffer(D._cabi.PEER_BLOB_BYTES)
        D._cabi.check(L.dlesm_halo_plan_peer_export(plan, 0, 1, blob))
        D._cabi.check(L.dlesm_halo_plan_peer_connect(plan, 0, 1, blob))
    if connect:
        assert L.dlesm_halo_plan_peer_connected(plan) == 1
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    return L, g, x, y, plan, oc


@pytest.mark.parametrize("nx,ny,alignment", [(300, 41, 64), (64, 64, 2), (1500, 700, 64), (130, 5, 2), (37, 29, None),
                                             (3, 3, 64), (8, 3, None), (2048, 2048, 64), (2049, 1031, None)])
@pytest.mark.parametrize("connect,join_inside", [("rccl", 1), ("host", 1), ("host", 0)])
def test_joined_steps_over_the_mailboxes(D, nx, ny, alignment, connect, join_inside):
    """dlesm_jacobi5_step_dm on a connected plan: after every step the output (edge halos included) equals the oracle's
    stencil + edge exchange; odd leading dimensions and boxes without an interior take the frame-only launch.
    join_inside (dm_peer_join_fused, default 1): the join rides in the step's launch (a few workgroups behind the frame
    workgroups wait for this step's strips and copy them into the halos); 0: the separate wait + unpack launch"""
    import torch
    L, g, x, y, plan, oc = _setup(D, nx, ny, alignment, connect)
    L.dlesm_set_tuning(b"dm_peer_join_fused", join_inside)
    it = x.internal
    D.psy.hash_init(x, SEED + 31)
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, None))
    D.copy_field(x, y)
    for _ in range(5):
        hx, want = x.get_data(), y.get_data()
        O.jacobi5(hx, want, g.nx, *it.box())
        assert O.exchange_dirs([want], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
        D._cabi.check(L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *it.box(), None))
        torch.cuda.synchronize()
        assert np.array_equal(y.get_data(), want)
        x, y = y, x
    L.dlesm_set_tuning(b"dm_peer_join_fused", 1)
    assert L.dlesm_wait_timed_out(0) == 0
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))


@pytest.mark.parametrize("nx,ny,alignment,nsteps", [(300, 41, 64, 7), (64, 64, 2, 12), (1500, 700, 64, 9), (130, 5, 2, 4),
                                                    (37, 29, None, 6), (3, 3, 64, 5), (4096, 4096, 64, 6), (2049, 1031, None, 5)])
@pytest.mark.parametrize("chain", [1, 0])
def test_time_loop_over_the_mailboxes(D, nx, ny, alignment, nsteps, chain):
    """dlesm_jacobi5_step_dm_pipelined on a connected plan: one launch per step, the frame workgroups wait for the
    neighbours' arrival flags and read their halo operands in the mailbox (parity of the previous step); ONE join at the
    end unpacks.  Then the mixtures: an RCCL exchange behind a pending step, a joined step behind a pipelined one, dm_peer=0
    (back to RCCL) and on again -- the sequence numbers of the two transports are independent"""
    import torch
    L, g, x, y, plan, oc = _setup(D, nx, ny, alignment)
    L.dlesm_set_tuning(b"j5_dm_chain", chain)
    it = x.internal
    D.psy.hash_init(x, SEED + 13)
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, None))
    D.copy_field(x, y)
    torch.cuda.synchronize()
    hx, hy = x.get_data(), y.get_data()
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    a, b = x, y

    def oracle_step():
        nonlocal hx, hy
        O.jacobi5(hx, hy, g.nx, *it.box())
        assert O.exchange_dirs([hy], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
        hx, hy = hy, hx

    try:
        for _ in range(nsteps):
            D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
            a, b = b, a
            oracle_step()
        D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
        s.synchronize()
        assert np.array_equal(a.get_data(), hx)
        # a plain (RCCL) exchange after a pending peer step joins by itself
        D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
        D._cabi.check(L.dlesm_halo_exchange_f64(plan, b.device_ptr, D._cabi.DIRS_ALL, sp))
        s.synchronize()
        O.jacobi5(hx, hy, g.nx, *it.box())
        assert O.exchange_all([hy], [g.nx], [oc]) == 0
        assert np.array_equal(b.get_data(), hy)
        hx, hy = hy, hx
        a, b = b, a
        # a joined step behind a pipelined one; then the RCCL transport for two steps, then the mailboxes again
        D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
        D._cabi.check(L.dlesm_jacobi5_step_dm(plan, b.device_ptr, a.device_ptr, g.nx, g.ny, *it.box(), sp))
        oracle_step(), oracle_step()
        s.synchronize()
        assert np.array_equal(a.get_data(), hx)
        L.dlesm_set_tuning(b"dm_peer", 0)
        for _ in range(2):
            D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
            a, b = b, a
            oracle_step()
        L.dlesm_set_tuning(b"dm_peer", 1)
        for _ in range(3):
            D._cabi.check(L.dlesm_jacobi5_step_dm_pipelined(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
            a, b = b, a
            oracle_step()
        # joining on ANOTHER stream than the one the steps ran on
        D._cabi.check(L.dlesm_halo_plan_join(plan, None))
        torch.cuda.synchronize()
        assert np.array_equal(a.get_data(), hx)
        assert L.dlesm_wait_timed_out(0) == 0
    finally:
        L.dlesm_set_tuning(b"dm_peer", 1)
        L.dlesm_set_tuning(b"j5_dm_chain", 1)
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))


def test_eight_direction_steps_and_refusals(D):
    """j5_dm_corners=1: the four corner messages travel through the mailboxes too (halos then equal a full exchange);
    what the transport does not take is refused in words: a second connect, a mailbox for another field count
    (captures into a graph are taken since the sequence numbers moved to the device: test_mailbox_steps_captured_into_a_graph)"""
    import torch
    L, g, x, y, plan, oc = _setup(D, 257, 63, 64)
    it = x.internal
    L.dlesm_set_tuning(b"j5_dm_corners", 1)
    try:
        D.psy.hash_init(x, SEED + 5)
        D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, None))
        D.copy_field(x, y)
        for _ in range(4):
            hx, want = x.get_data(), y.get_data()
            O.jacobi5(hx, want, g.nx, *it.box())
            assert O.exchange_all([want], [g.nx], [oc]) == 0
            D._cabi.check(L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *it.box(), None))
            torch.cuda.synchronize()
            assert np.array_equal(y.get_data(), want)
            x, y = y, x
    finally:
        L.dlesm_set_tuning(b"j5_dm_corners", 0)
    blob = C.create_string_buffer(D._cabi.PEER_BLOB_BYTES)
    assert L.dlesm_halo_plan_peer_export(plan, 0, 1, blob) != 0 and "already connected" in D._cabi.last_error()
    assert L.dlesm_halo_plan_peer_connect_rccl(plan, 1) != 0
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))
    # a blob that does not describe the receive this rank's send expects
    L, g, x, y, plan, oc = _setup(D, 40, 30, 64, connect=None)
    D._cabi.check(L.dlesm_halo_plan_peer_export(plan, 0, 1, blob))
    bad = C.create_string_buffer(blob.raw, D._cabi.PEER_BLOB_BYTES)
    bad[0] = b"X"
    assert L.dlesm_halo_plan_peer_connect(plan, 0, 1, bad) != 0 and "not valid" in D._cabi.last_error()
    two = C.create_string_buffer(blob.raw + blob.raw, 2 * D._cabi.PEER_BLOB_BYTES)
    assert L.dlesm_halo_plan_peer_connect(plan, 1, 2, two) != 0      # rank 1 of 2: nobody sends to rank 1's slots
    assert L.dlesm_halo_plan_peer_connected(plan) == 0
    D._cabi.check(L.dlesm_halo_plan_peer_connect(plan, 0, 1, blob))
    assert L.dlesm_halo_plan_peer_connected(plan) == 1
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))


@pytest.mark.parametrize("dirs,no_diag", [((), False), ((1,), False), ((1, 4), False), ((1, 2, 3), False), ((1, 2, 3, 4), False),
                                          ((1, 2, 3, 4), True), ((2, 4), True)])
@pytest.mark.parametrize("nx,ny,alignment,nf,one_launch", [(37, 23, 8, 1, 0), (130, 6, None, 3, 0), (700, 300, 64, 2, 0),
                                                           (37, 23, 8, 1, 1), (700, 300, 64, 3, 1)])
def test_halo_exchange_over_the_mailboxes(D, nx, ny, alignment, nf, one_launch, dirs, no_diag):
    """r2d_field%halo_exchange (exchange_generic with a choice of comm1..comm4, parallel_comms_mod.f90:1557-1571) on a
    connected plan: two small launches, no RCCL.  dm_skip_parts=1 switches the RCCL group OFF for the duration, so only
    the mailbox path can produce the oracle's halos; then the same exchange through RCCL (dm_peer_exchange=0) and through
    the mailboxes again -- the two transports keep separate sequence numbers"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    g = _grid(D, nx, ny, alignment)
    F = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(nf)]
    t = loopback_tables(D, F[0].internal)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    D._cabi.check(L.dlesm_halo_plan_peer_connect_rccl(plan, 3))
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    mask = sum(1 << (d - 1) for d in dirs) | (D._cabi.DIRS_NO_DIAGONALS if no_diag else 0)
    ptrs = (C.c_void_p * nf)(*[f.device_ptr for f in F])
    L.dlesm_set_tuning(b"dm_peer_one_launch", one_launch)      # 1 (default): both halves of the exchange in ONE launch
    try:
        for rnd, (skip, peer_x) in enumerate(((1, 1), (0, 0), (1, 1), (1, 1))):
            for k, f in enumerate(F):
                D.psy.hash_init(f, SEED + 100 * rnd + k)
            torch.cuda.synchronize()
            want = [f.get_data() for f in F]
            for w in want:
                assert O.exchange_dirs([w], [g.nx], [oc], dirs, no_diagonals=no_diag) == 0
            L.dlesm_set_tuning(b"dm_skip_parts", skip)
            L.dlesm_set_tuning(b"dm_peer_exchange", peer_x)
            D._cabi.check(L.dlesm_halo_exchange_multi_f64(plan, ptrs, nf, mask, None))
            torch.cuda.synchronize()
            for f, w in zip(F, want):
                assert np.array_equal(f.get_data(), w), (rnd, skip, peer_x)
        assert L.dlesm_wait_timed_out(0) == 0
    finally:
        L.dlesm_set_tuning(b"dm_skip_parts", 0)
        L.dlesm_set_tuning(b"dm_peer_exchange", 1)
        L.dlesm_set_tuning(b"dm_peer_one_launch", 1)
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))


def test_deep_halo_exchange_over_the_mailboxes(D):
    """a depth-3 plan (the tables of the fused multi-step forms): strips three cells deep, 3 x 3 corners -- the mailbox
    exchange has no one-cell-ring condition"""
    import torch
    from dm_overhead import loopback_tables
    L = D._cabi.lib()
    os.environ["DL_ESM_ALIGNMENT"] = "64"
    g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
    g.decompose(200, 90, halo_width=3)
    D.grid_init(g, 1.0, 1.0)
    os.environ.pop("DL_ESM_ALIGNMENT", None)
    f = D.r2d_field(g, D.GO_T_POINTS)
    t = loopback_tables(D, f.internal, 3)
    plan = C.c_void_p()
    D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
    D._cabi.check(L.dlesm_halo_plan_peer_connect_rccl(plan, 1))
    oc = O.Comms()
    C.memmove(C.byref(oc), C.byref(t), C.sizeof(oc))
    L.dlesm_set_tuning(b"dm_skip_parts", 1)
    try:
        for rnd in range(3):
            D.psy.hash_init(f, SEED + rnd, box=D._cabi.Region(0, 0, 1, g.nx, 1, g.ny))
            torch.cuda.synchronize()
            want = f.get_data()
            assert O.exchange_all([want], [g.nx], [oc]) == 0
            D._cabi.check(L.dlesm_halo_exchange_f64(plan, f.device_ptr, D._cabi.DIRS_ALL, None))
            torch.cuda.synchronize()
            assert np.array_equal(f.get_data(), want)
    finally:
        L.dlesm_set_tuning(b"dm_skip_parts", 0)
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))


def test_mailbox_operations_issued_on_alternating_streams(D):
    """the mailboxes, their counter and the sequence number are one resource per plan: exchanges and steps issued on two
    non-blocking streams in turn, with no host synchronisation in between, are ordered by the library (peer_order)"""
    import torch
    L, g, x, y, plan, oc = _setup(D, 1500, 700, 64)
    it = x.internal
    D.psy.hash_init(x, SEED + 77)
    D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, None))
    D.copy_field(x, y)
    torch.cuda.synchronize()
    hx, hy = x.get_data(), y.get_data()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a, b = x, y
    ev = torch.cuda.Event()
    for k in range(12):
        s = (s1, s2)[k % 2]
        o = (s2, s1)[k % 2]
        s.wait_event(ev)                       # the FIELDS are the caller's to order; the mailboxes are the library's
        sp = C.c_void_p(s.cuda_stream)
        if k % 3 == 2:
            D._cabi.check(L.dlesm_halo_exchange_f64(plan, a.device_ptr, D._cabi.DIRS_ALL, sp))
            assert O.exchange_all([hx], [g.nx], [oc]) == 0
        else:
            fn = L.dlesm_jacobi5_step_dm_pipelined if k % 3 == 0 else L.dlesm_jacobi5_step_dm
            D._cabi.check(fn(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), sp))
            a, b = b, a
            O.jacobi5(hx, hy, g.nx, *it.box())
            assert O.exchange_dirs([hy], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
            hx, hy = hy, hx
        ev = torch.cuda.Event()
        ev.record(s)
        del o
    D._cabi.check(L.dlesm_halo_plan_join(plan, None))
    torch.cuda.synchronize()
    assert np.array_equal(a.get_data(), hx)
    assert L.dlesm_wait_timed_out(0) == 0
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))


@pytest.mark.parametrize("nx,ny,alignment,nsteps", [(300, 41, 64, 5), (37, 29, None, 4), (1500, 700, 64, 6)])
def test_fenced_mailboxes_change_no_bit(D, nx, ny, alignment, nsteps):
    """mailbox_fences = 1: the arrival flags are system-scope RELEASE stores and the waits end in an ACQUIRE fence (the
    form bench.py falls back to if the default one fails its self-check between GPUs): same bits, every entry that uses
    the mailboxes -- time loop, joined step, the exchange in one and in two launches"""
    import torch
    L, g, x, y, plan, oc = _setup(D, nx, ny, alignment)
    it = x.internal
    L.dlesm_set_tuning(b"mailbox_fences", 1)
    try:
        D.psy.hash_init(x, SEED + 41)
        want = x.get_data()
        assert O.exchange_all([want], [g.nx], [oc]) == 0
        for one_launch in (1, 0):                # the exchange on its own, both forms (the second one changes nothing)
            L.dlesm_set_tuning(b"dm_peer_one_launch", one_launch)
            D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, None))
            assert np.array_equal(x.get_data(), want)
        L.dlesm_set_tuning(b"dm_peer_one_launch", 1)
        D.copy_field(x, y)
        torch.cuda.synchronize()
        hx, hy = x.get_data(), y.get_data()
        a, b = x, y
        for k in range(nsteps):
            fn = L.dlesm_jacobi5_step_dm if k == nsteps // 2 else L.dlesm_jacobi5_step_dm_pipelined
            D._cabi.check(fn(plan, a.device_ptr, b.device_ptr, g.nx, g.ny, *it.box(), None))
            a, b = b, a
            O.jacobi5(hx, hy, g.nx, *it.box())
            assert O.exchange_dirs([hy], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0
            hx, hy = hy, hx
        D._cabi.check(L.dlesm_halo_plan_join(plan, None))
        torch.cuda.synchronize()
        assert np.array_equal(a.get_data(), hx)
        assert L.dlesm_wait_timed_out(0) == 0
    finally:
        L.dlesm_set_tuning(b"mailbox_fences", 0)
        L.dlesm_set_tuning(b"dm_peer_one_launch", 1)
        D._cabi.check(L.dlesm_halo_plan_destroy(plan))


# --------------------------------------------------------------------------- hipGraph capture of mailbox operations
@pytest.mark.parametrize("form", ["joined", "time_loop", "exchange", "joined_apart"])
@pytest.mark.parametrize("nx,ny,alignment", [(300, 41, 64), (37, 29, None), (1500, 700, 64)])
def test_mailbox_steps_captured_into_a_graph(D, form, nx, ny, alignment):
    """The sequence numbers of the mailboxes live on the device (peer_seq_load), so mailbox operations can be
    captured: a graph of TWO ping-pong steps -- joined form, time-loop form + join, plain sweep + halo_exchange over the
    mailboxes, joined form with the join as its own launch -- replayed three times, with un-captured steps before, between
    and after the replays, equals the oracle's steps + edge exchanges bit for bit.  (No RCCL call is in the capture: this
    works with the RCCL that cannot be captured, too.)"""
    import torch
    L, g, x, y, plan, oc = _setup(D, nx, ny, alignment, "host")
    it = x.internal
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    L.dlesm_set_tuning(b"dm_peer_join_fused", 0 if form == "joined_apart" else 1)

    def issue(src, dst):
        if form in ("joined", "joined_apart"):
            return [L.dlesm_jacobi5_step_dm(plan, src.device_ptr, dst.device_ptr, g.nx, g.ny, *it.box(), sp)]
        if form == "time_loop":
            return [L.dlesm_jacobi5_step_dm_pipelined(plan, src.device_ptr, dst.device_ptr, g.nx, g.ny, *it.box(), sp)]
        return [L.dlesm_stencil5_f64(src.device_ptr, dst.device_ptr, g.nx, g.ny, *it.box(), sp),
                L.dlesm_halo_exchange_f64(plan, dst.device_ptr, D._cabi.DIRS_EDGES_ONLY, sp)]

    try:
        with torch.cuda.stream(s):
            D.psy.hash_init(x, SEED + 91, stream=s)
            D._cabi.check(L.dlesm_halo_exchange_f64(plan, x.device_ptr, D._cabi.DIRS_ALL, sp))    # (the capturing stream: see peer_order)
            D.copy_field(x, y, stream=s)
        s.synchronize()
        hx, hy = x.get_data(), y.get_data()

        def oracle_step(src, dst):
            O.jacobi5(src, dst, g.nx, *it.box())
            assert O.exchange_dirs([dst], [g.nx], [oc], (1, 2, 3, 4), no_diagonals=True) == 0

        def eager_pair():
            assert all(rc == 0 for rc in issue(x, y) + issue(y, x)), L.dlesm_last_error()
            D._cabi.check(L.dlesm_halo_plan_join(plan, sp))
            oracle_step(hx, hy)
            oracle_step(hy, hx)

        eager_pair()                         # three operations so far (exchange + 2): the graph starts on an even parity
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            rcs = issue(x, y) + issue(y, x) + [L.dlesm_halo_plan_join(plan, sp)]
        assert all(rc == 0 for rc in rcs), L.dlesm_last_error()
        for k in range(3):
            with torch.cuda.stream(s):            # (replay() launches on the current stream)
                graph.replay()
            oracle_step(hx, hy)
            oracle_step(hy, hx)
            if k == 1:
                s.synchronize()
                eager_pair()                 # un-captured operations between replays: an even number keeps the parities
        eager_pair()
        torch.cuda.synchronize()
        assert L.dlesm_wait_timed_out(0) == 0
        assert np.array_equal(x.get_data(), hx)
        if form == "time_loop":              # the halos of the last-but-one level were read from the mailbox, never unpacked into the field
            inner = (slice(it.ystart - 1, it.ystop), slice(it.xstart - 1, it.xstop))
            assert np.array_equal(y.get_data()[inner], hy[inner])
        else:
            assert np.array_equal(y.get_data(), hy)
        del graph
    finally:
        L.dlesm_set_tuning(b"dm_peer_join_fused", 1)
    D._cabi.check(L.dlesm_halo_plan_destroy(plan))


def test_a_graph_replayed_out_of_step_is_reported(D):
    """a graph that holds an ODD number of mailbox operations has the wrong mailbox halves baked in from its second replay on:
    the workgroup that raises the flags finds the last raised number != its own - 1 and raises the process-wide sticky word
    (every later entry then fails, as after a time-out) -- wrong halos cannot leave silently"""
    import torch
    L, g, x, y, plan, oc = _setup(D, 130, 40, 64, "host")
    it = x.internal
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    try:
        with torch.cuda.stream(s):
            D.psy.hash_init(x, SEED + 92, stream=s)
            D.copy_field(x, y, stream=s)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            rc = L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *it.box(), sp)
        assert rc == 0, L.dlesm_last_error()
        with torch.cuda.stream(s):            # (replay() launches on the current stream)
            graph.replay()
        torch.cuda.synchronize()
        assert L.dlesm_wait_timed_out(0) == 0            # the first replay is in step
        with torch.cuda.stream(s):            # (replay() launches on the current stream)
            graph.replay()
        torch.cuda.synchronize()
        assert L.dlesm_wait_timed_out(0) == 1
        assert L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *it.box(), sp) != 0
        del graph
    finally:
        L.dlesm_halo_plan_destroy(plan)
        L.dlesm_wait_timed_out(1)
    assert L.dlesm_wait_timed_out(0) == 0
