"""One rank of the CPU multi-process tests (launched by tests/test_multirank_gloo.py, gloo
backend, no GPU).  It drives the PRODUCT's host logic for its own rank -- parallel_init,
grid_type%decompose, grid_init (-> map_comms through the C ABI) -- and then plays the
reference's dist_mem tests (test_halos / test_gsum / test_reduction) with numpy arrays as the
fields and gloo send/recv as a TEST-ONLY transport that issues exactly the calls the product's plan reports
(dlesm_halo_plan_describe: peer, direction, count, staging slot, in ISSUE ORDER) and, like RCCL, uses NO tags:
messages between a pair of ranks match purely by issue order, so the C++ ordering / masking / slot logic of
dlesm_halo.hip itself is what is tested.  What this checks is
everything about the N>1 path that is not the GPU itself: tile ownership, message tables,
peer/ordering logic, scatter/gather index maps.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/gloo_worker.py NX NY
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ref_cases as R  # noqa: E402


DIRS_ALL, DIRS_NO_DIAGONALS = 0xF, 0x10


def plan_calls(tables, ld, ny, nf, mask, aggregated):
    """the ncclRecv / ncclSend calls of ONE exchange, in issue order, as the PRODUCT reports them
    (dlesm_halo_plan_describe: built by the code dlesm_halo_plan_create uses, walked by the code the exchanges use)"""
    import ctypes as C
    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    n = C.c_int()
    D._cabi.check(L.dlesm_halo_plan_describe(C.byref(tables), ld, ny, nf, mask, 1 if aggregated else 0, None, 0, C.byref(n)))
    out = (D._cabi.MsgDesc * max(1, n.value))()
    D._cabi.check(L.dlesm_halo_plan_describe(C.byref(tables), ld, ny, nf, mask, 1 if aggregated else 0, out, n.value, C.byref(n)))
    return [out[k] for k in range(n.value)]


def exchange_multi(fields, tables, rank, mask=DIRS_ALL, aggregated=True):
    """halo exchange of several (ny, ld) numpy fields over gloo, issued from the product's OWN call list: one
    irecv / isend per reported ncclRecv / ncclSend, in the reported order, with NO tags (every message carries tag 0,
    as RCCL has none) -- between a pair of ranks the k-th send therefore has to meet the k-th receive.  Payloads go
    through staging buffers at the reported offsets (overlapping slots would corrupt them); in-place messages go
    straight from / to the field.  A wrong sort, a wrong skip of a masked direction or a wrong slot layout in the
    C++ shows up here as a mismatched or corrupted message."""
    ny, ld = fields[0].shape
    nf = len(fields)
    calls = plan_calls(tables, ld, ny, nf, mask, aggregated)
    span = {True: 0, False: 0}
    for c in calls:
        if c.buffer_offset >= 0:
            span[bool(c.is_recv)] = max(span[bool(c.is_recv)], c.buffer_offset + c.count)
    rbuf, sbuf = torch.full((span[True],), float("nan"), dtype=torch.float64), torch.full((span[False],), float("nan"),
                                                                                          dtype=torch.float64)
    used = {True: [], False: []}

    def slot(c):
        """the call's slice of its staging buffer; slots of one exchange must not overlap"""
        lo, hi = c.buffer_offset, c.buffer_offset + c.count
        for (a, b) in used[bool(c.is_recv)]:
            assert hi <= a or b <= lo, f"rank {rank}: staging slots overlap: [{lo},{hi}) and [{a},{b})"
        used[bool(c.is_recv)].append((lo, hi))
        return (rbuf if c.is_recv else sbuf)[lo:hi]

    def strip(f, c):
        return f[c.j0 - 1:c.j0 - 1 + c.ny, c.i0 - 1:c.i0 - 1 + c.nx]

    reqs, landed = [], []
    for c in calls:                                            # THE issue order
        which = list(range(nf)) if c.field < 0 else [c.field]
        assert c.count == len(which) * c.nx * c.ny
        if c.is_recv:
            buf = slot(c) if c.buffer_offset >= 0 else torch.empty(c.count, dtype=torch.float64)
            reqs.append(dist.irecv(buf, src=c.peer, tag=0))
            landed.append((c, which, buf))
        else:
            payload = torch.from_numpy(np.concatenate([np.ascontiguousarray(strip(fields[k], c)).reshape(-1) for k in which]))
            if c.buffer_offset >= 0:
                slot(c).copy_(payload)
                payload = slot_view = sbuf[c.buffer_offset:c.buffer_offset + c.count]   # noqa: F841  (sent from the slot)
            reqs.append(dist.isend(payload, dst=c.peer, tag=0))
    for q in reqs:
        q.wait()
    for c, which, buf in landed:
        got = buf.numpy().reshape(len(which), c.ny, c.nx)
        for n, k in enumerate(which):
            strip(fields[k], c)[...] = got[n]


def exchange(field, tables, rank, mask=DIRS_ALL, aggregated=True):
    exchange_multi([field], tables, rank, mask, aggregated)


def deep_halo_suite(D, nx, ny, depth, rank, world):
    """The depth-d extension (dlesm_map_comms_depth) and the algorithm of the fused distributed
    step: (1) after a depth-d exchange every halo cell within d of the tile that lies inside the
    domain holds the owner's value; (2) d staged oracle steps on the tile -- stage boxes grown
    by d-s cells towards every neighbour, as dlesm_jacobi5_multi_step_dm does -- reproduce, bit
    for bit, d oracle steps on the undecomposed domain."""
    import oracle_lib as O
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(nx, ny, halo_width=depth)
    D.grid_init(g, 1.0, 1.0)
    sub, t = g.subdomain, g.comm_tables
    internal, _ = D.field_mod.field_bounds(g, R.GO_T)
    xs, xe, ys, ye = internal.box()
    assert xs == depth + 1 and ys == depth + 1
    G = np.random.default_rng(1234).random((ny + 2, nx + 2))   # [gj, gi], ring at 0 and n+1
    gx = lambda i: sub.glob.xstart + (i - xs)                  # noqa: E731  local -> global, 1-based
    gy = lambda j: sub.glob.ystart + (j - ys)                  # noqa: E731
    f = np.zeros((g.ny, g.nx))
    for j in range(ys - 1, ye + 2):
        for i in range(xs - 1, xe + 2):
            inside = xs <= i <= xe and ys <= j <= ye
            on_ring = gx(i) in (0, nx + 1) or gy(j) in (0, ny + 1)
            if inside or on_ring:
                f[j - 1, i - 1] = G[gy(j), gx(i)]
    exchange(f, t, rank)
    errors = 0
    for j in range(ys - depth, ye + depth + 1):
        for i in range(xs - depth, xe + depth + 1):
            if 1 <= gx(i) <= nx and 1 <= gy(j) <= ny and f[j - 1, i - 1] != G[gy(j), gx(i)]:
                if errors < 3:
                    print(f"rank {rank}: ERROR depth-{depth} halo cell ({i},{j}) = {f[j - 1, i - 1]}", flush=True)
                errors += 1
    # (2) the staged steps against the undecomposed domain
    want, tmp = G.copy(), G.copy()
    for _ in range(depth):
        O.jacobi5(want, tmp, nx + 2, 2, nx + 1, 2, ny + 1)
        want, tmp = tmp, want
        tmp[:] = want                                          # same ring in both buffers
    hasW, hasE = int(sub.glob.xstart > 1), int(sub.glob.xstop < nx)
    hasS, hasN = int(sub.glob.ystart > 1), int(sub.glob.ystop < ny)
    cur = f
    for s in range(1, depth + 1):
        k = depth - s
        nxt = cur.copy()
        O.jacobi5(cur, nxt, g.nx, xs - hasW * k, xe + hasE * k, ys - hasS * k, ye + hasN * k)
        cur = nxt
    got = cur[ys - 1:ye, xs - 1:xe]
    ref = want[sub.glob.ystart:sub.glob.ystop + 1, sub.glob.xstart:sub.glob.xstop + 1]
    if not np.array_equal(got, ref):
        print(f"rank {rank}: ERROR staged {depth}-step result differs from the undecomposed domain", flush=True)
        errors += 1
    return errors


def main():
    nx, ny = int(sys.argv[1]), int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dl_esm_inf_amd as D
    D.parallel_init(rank, world, use_rccl=False)
    assert D.get_rank() == rank + 1 and D.get_num_ranks() == world
    if len(sys.argv) > 3:
        tot = torch.tensor([deep_halo_suite(D, nx, ny, int(sys.argv[3]), rank, world)])
        dist.all_reduce(tot)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(1 if int(tot[0]) else 0)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(nx, ny)
    D.grid_init(g, 1.0, 1.0)
    sub = g.subdomain
    t = g.comm_tables
    # every message has a partner on the other side: exchange the table summaries
    mine = [(m["dest"], m["dir"], m["nx"] * m["ny"]) for m in t.sends()]
    everyone = [None] * world
    dist.all_gather_object(everyone, {"sends": mine, "recvs": [(m["src"], m["dir"], m["nx"] * m["ny"])
                                                              for m in t.recvs()]})
    for peer, summary in enumerate(everyone):
        for (dest, d, n) in summary["sends"]:
            if dest == rank:
                assert (peer, d, n) in [(m["src"], m["dir"], m["nx"] * m["ny"]) for m in t.recvs()]

    errors = 0
    # ---- test_halos: U, V, T, F fields --------------------------------------------------
    for ptype in (R.GO_T, R.GO_U, R.GO_V, R.GO_F):
        internal, _ = D.field_mod.field_bounds(g, ptype)
        it = internal.box()
        f = R.init_field_hill(ptype, g.nx, g.ny, it, sub.glob.xstart, sub.glob.ystart)
        before = f.copy()
        f2 = f.copy()
        exchange(f, t, rank)                                    # the aggregated form (dlesm_halo_exchange_f64 on its own)
        exchange(f2, t, rank, aggregated=False)                 # rows in place, strided strips through the pack buffer
        bad = R.check_hill_halos(f, ptype, it, sub.glob.box(), nx, ny, corners=True)
        if bad or not np.array_equal(f, f2):
            print(f"rank {rank}: ERROR in halo values for ptype {ptype}: {bad[:3]} (forms equal: {np.array_equal(f, f2)})", flush=True)
            errors += 1
        xs, xe, ys, ye = it
        if not np.array_equal(f[ys - 1:ye, xs - 1:xe], before[ys - 1:ye, xs - 1:xe]):
            print(f"rank {rank}: ERROR internal cells modified", flush=True)
            errors += 1
    # ---- three fields in ONE grouped exchange (what dlesm_shallow_step_dm issues), tag-free ------------
    fields, befores = [], []
    for k, ptype in enumerate((R.GO_U, R.GO_V, R.GO_T)):
        internal, _ = D.field_mod.field_bounds(g, ptype)
        f = R.init_field_hill(ptype, g.nx, g.ny, internal.box(), sub.glob.xstart, sub.glob.ystart)
        f *= (k + 1)                                           # a mixed-up field order would show
        fields.append(f)
    exchange_multi(fields, t, rank)
    for k, ptype in enumerate((R.GO_U, R.GO_V, R.GO_T)):
        internal, _ = D.field_mod.field_bounds(g, ptype)
        bad = R.check_hill_halos(fields[k] / (k + 1), ptype, internal.box(), sub.glob.box(), nx, ny, corners=True)
        if bad:
            print(f"rank {rank}: ERROR grouped exchange, field {k}: {bad[:3]}", flush=True)
            errors += 1
    # ---- edges only (what dlesm_jacobi5_step_dm exchanges): corner halos stay untouched ----------------
    internal, _ = D.field_mod.field_bounds(g, R.GO_T)
    it = internal.box()
    f = R.init_field_hill(R.GO_T, g.nx, g.ny, it, sub.glob.xstart, sub.glob.ystart)
    xs, xe, ys, ye = it
    corners = [(ys - 2, xs - 2), (ys - 2, xe), (ye, xs - 2), (ye, xe)]
    for (j, i) in corners:
        f[j, i] = -123.0
    exchange(f, t, rank, DIRS_ALL | DIRS_NO_DIAGONALS, aggregated=False)     # the Jacobi step's own form and mask
    if any(f[j, i] != -123.0 for (j, i) in corners):
        print(f"rank {rank}: ERROR edges-only exchange touched a corner halo", flush=True)
        errors += 1
    bad = R.check_hill_halos(f, R.GO_T, it, sub.glob.box(), nx, ny, corners=False)
    if bad:
        print(f"rank {rank}: ERROR edges-only exchange: {bad[:3]}", flush=True)
        errors += 1
    # ---- comm1..comm4 subsets (exchange_generic's masks): both forms issue the same, consistent, subset -------------
    for mask in (0xF, 0x1, 0x3, 0xC, 0x5, 0xA, 0x0, 0xF | DIRS_NO_DIAGONALS):
        res = []
        for agg in (True, False):
            fm = [R.init_field_hill(R.GO_T, g.nx, g.ny, it, sub.glob.xstart, sub.glob.ystart) * (k + 1) for k in range(2)]
            for f in fm:                                           # halos start wrecked: what arrives is what is compared
                keep = f[ys - 1:ye, xs - 1:xe].copy()
                f[:] = -77.0
                f[ys - 1:ye, xs - 1:xe] = keep
            exchange_multi(fm, t, rank, mask, aggregated=agg)      # completes only if every send has its receive
            res.append(fm)
        # two fields that differ by a factor: a message delivered to the wrong field, or the two forms disagreeing, shows
        if not all(np.array_equal(a, b) for a, b in zip(*res)) or not np.array_equal(res[0][1], np.where(res[0][0] == -77.0, -77.0, 2 * res[0][0])):
            print(f"rank {rank}: ERROR masked exchange {mask:#x}: forms differ or fields mixed up", flush=True)
            errors += 1
    # ---- test_gsum: checksum of ones == jpiglo*jpjglo --------------------------------
    internal, _ = D.field_mod.field_bounds(g, R.GO_T)
    f = R.gsum_field(g.nx, g.ny, internal.box())
    xs, xe, ys, ye = internal.box()
    local = torch.tensor([np.abs(f[ys - 1:ye, xs - 1:xe]).sum()], dtype=torch.float64)
    dist.all_reduce(local)
    if float(local[0]) != float(nx * ny):
        print(f"rank {rank}: ERROR global sum {float(local[0])} != {nx * ny}", flush=True)
        errors += 1
    # ---- test_reduction: scatter by the subdomain's global box, gather back + 1 --------
    glob = R.unique_global(nx, ny)
    f = np.zeros((g.ny, g.nx))
    dx, dy = sub.glob.xstart - xs, sub.glob.ystart - ys
    f[ys - 1:ye, xs - 1:xe] = glob[ys + dy - 1:ye + dy, xs + dx - 1:xe + dx] + 1.0
    halo_x, halo_y = xs - 1, ys - 1
    n = (g.decomp.max_width - 2 * halo_x) * (g.decomp.max_height - 2 * halo_y)
    send = torch.zeros(n, dtype=torch.float64)
    inner = f[ys - 1:ye, xs - 1:xe].reshape(-1)
    send[:inner.size] = torch.from_numpy(inner.copy())
    slots = [torch.zeros(n, dtype=torch.float64) for _ in range(world)] if rank == 0 else None
    dist.gather(send, slots, dst=0)
    if rank == 0:
        back = np.zeros((ny, nx))
        for r in range(world):
            s = g.decomp.subdomains[r].glob
            w, h = s.xstop - s.xstart + 1, s.ystop - s.ystart + 1
            back[s.ystart - 1:s.ystop, s.xstart - 1:s.xstop] = slots[r].numpy()[:w * h].reshape(h, w)
        if not np.array_equal(back, glob + 1.0):
            print("rank 0: ERROR gathered field incorrect", flush=True)
            errors += 1
    tot = torch.tensor([errors])
    dist.all_reduce(tot)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if int(tot[0]) else 0)


if __name__ == "__main__":
    main()
