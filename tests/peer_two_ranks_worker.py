"""One rank of the N-process test of the PEER TRANSPORT on ONE GPU (tests/test_a_peer_two_ranks_gpu.py): the ranks are
separate processes that share device 0, so a neighbour's mailbox is reached through a real hipIpcMemHandle, arrival flags
are raised by ANOTHER process's kernel, and the ranks run skewed against each other -- everything the loop-back tests
cannot show except the xGMI hop itself.  No RCCL anywhere (it refuses two ranks on one device): the blobs travel through
a gloo group, the decomposition and the message tables are the product's own (go_decompose / map_comms).

Check: N time steps (joined form, then the time-loop form + one join) against the oracle's steps on the UNDIVIDED domain,
every internal cell and every edge halo of this rank's tile, bit for bit.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/peer_two_ranks_worker.py NX NY [STEPS]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NX, NY = int(sys.argv[1]), int(sys.argv[2])
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ALIGN = sys.argv[4] if len(sys.argv) > 4 else "64"
MODE = sys.argv[5] if len(sys.argv) > 5 else "connect"     # "mailbox": the library's mode without a communication library
SEED = 20261004 + 3
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo", rank=rank, world_size=world)
import dl_esm_inf_amd as D  # noqa: E402
import oracle_lib as O  # noqa: E402

torch.cuda.set_device(0)
L = D._cabi.lib()
L.dlesm_set_tuning(b"dm_wait_seconds", 30)         # a protocol error must end in words, not in a hung box
if MODE == "mailbox":
    D.parallel_init(rank, world, transport="mailbox")     # plans connect their own mailboxes (room for 3 fields)
else:
    D.parallel_init(rank, world, use_rccl=False)            # the host program connects them (halo_connect_peers below)
if ALIGN == "none":
    os.environ.pop("DL_ESM_ALIGNMENT", None)
else:
    os.environ["DL_ESM_ALIGNMENT"] = ALIGN
g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
g.decompose(NX, NY)
D.grid_init(g, 1.0, 1.0)
x, y = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
it = x.internal
sub = g.subdomain
gx0 = sub.glob.xstart - sub.internal.xstart + 1     # global index of local cell 1 (psy.hash_init)
gy0 = sub.glob.ystart - sub.internal.ystart + 1
ring = D._cabi.Region(0, 0, it.xstart - 1, it.xstop + 1, it.ystart - 1, it.ystop + 1)

# the undivided domain on the host: global cells 0 .. N+1 (0 and N+1 = the fixed boundary ring)
gld, gny = NX + 2, NY + 2
G = O.hash_field(SEED, gny, gld, 0, 0, 1, NX + 2, 1, NY + 2)
H = G.copy()
history = [G.copy()]
for _ in range(2 * STEPS + 4 + 4):       # joined steps, time-loop steps, two replays of a two-step graph, four steps of the masks test
    O.jacobi5(G, H, gld, 2, NX + 1, 2, NY + 1)
    G, H = H, G
    history.append(G.copy())


def check(fld, nsteps, what):
    got = fld.get_data()
    want = history[nsteps]
    bad = 0
    # internal region + the four edge halos (a 5-point step exchanges no corner)
    for (x0, x1, y0, y1) in ((it.xstart, it.xstop, it.ystart - 1, it.ystop + 1), (it.xstart - 1, it.xstop + 1, it.ystart, it.ystop)):
        loc = got[y0 - 1:y1, x0 - 1:x1]
        glo = want[gy0 + y0 - 1:gy0 + y1, gx0 + x0 - 1:gx0 + x1]
        bad += int(np.count_nonzero(loc != glo))
    if bad:
        print(f"ERROR rank {rank}: {what}: {bad} cells differ from the undivided oracle after {nsteps} steps", flush=True)
    return bad


D.psy.halo_connect_peers(g)            # (mailbox mode: already connected when the plan was made -- a no-op)
errors = 0
# r2d_field%halo_exchange between the processes (no RCCL in this job: only the mailboxes can do it).  y: a WRONG field
# everywhere, the right one on the internal region; after the exchange every halo cell that lies inside the global domain
# (edges and corners) must hold the right field, the cells of the global boundary ring keep the wrong one.
for rnd in range(3):
    D.psy.hash_init(y, SEED + 1 + rnd, box=ring)
    D.psy.hash_init(y, SEED + 50 + rnd, box=it)
    y.halo_exchange(1)
    torch.cuda.synchronize()
    got = y.get_data()[it.ystart - 2:it.ystop + 1, it.xstart - 2:it.xstop + 1]
    right = O.hash_field(SEED + 50 + rnd, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
    wrong = O.hash_field(SEED + 1 + rnd, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
    want = wrong.copy()
    want[1:NY + 1, 1:NX + 1] = right[1:NY + 1, 1:NX + 1]
    want = want[gy0 + it.ystart - 2:gy0 + it.ystop + 1, gx0 + it.xstart - 2:gx0 + it.xstop + 1]
    bad = int(np.count_nonzero(got != want))
    if bad:
        print(f"ERROR rank {rank}: halo_exchange over the mailboxes, round {rnd}: {bad} cells differ", flush=True)
        errors += bad
D.psy.hash_init(x, SEED, box=ring)
D.psy.hash_init(y, SEED, box=ring)
s = torch.cuda.Stream()
a, b = x, y
n = 0
# joined form
for k in range(STEPS):
    D.psy.invoke_jacobi5_dm(b, a, stream=s)
    a, b = b, a
    n += 1
    if k in (0, STEPS - 1):
        s.synchronize()
        errors += check(a, n, "joined step")
# time-loop form: ranks deliberately skewed (rank r sleeps r x 50 ms in the middle)
for k in range(STEPS):
    D.psy.invoke_jacobi5_dm_pipelined(b, a, stream=s)
    a, b = b, a
    n += 1
    if k == STEPS // 2:
        s.synchronize()
        import time
        time.sleep(0.05 * rank)
D.psy.halo_join(g, stream=s)
s.synchronize()
errors += check(a, n, "time-loop form")
# the same two time-loop steps + join captured into ONE hipGraph per rank and replayed twice: the sequence numbers of the
# mailboxes live on the device and advance with every replay, on every rank alike (no communication library call in the graph)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
    D.psy.invoke_jacobi5_dm_pipelined(b, a, stream=s)
    D.psy.invoke_jacobi5_dm_pipelined(a, b, stream=s)
    D.psy.halo_join(g, stream=s)
for k in range(2):
    with torch.cuda.stream(s):            # (replay() launches on the current stream)
        graph.replay()
    n += 2
    if k == 0:
        import time
        time.sleep(0.03 * rank)
torch.cuda.synchronize()
errors += check(a, n, "time-loop steps replayed from a graph")
del graph
if MODE == "mailbox":
    # ---- everything else a multi-rank job needs, with no communication library underneath -------------------------
    # (1) field_checksum: the local sums travel over the host-side board and are added in rank order
    cs = D.field_checksum(a)
    want_cs = float(np.abs(history[n][1:NY + 1, 1:NX + 1]).sum())
    if abs(cs - want_cs) > 1e-11 * want_cs:
        print(f"ERROR rank {rank}: field_checksum {cs!r}, the undivided field gives {want_cs!r}", flush=True)
        errors += 1
    # (2) gather_inner_data: every rank copies its block straight into the root's buffer through an IPC mapping
    #     (round 4: into a gather buffer the library owns on the root, mapped ONCE per job by every other rank -- the
    #     second gather, of the previous time level, goes through the mapping the first one opened)
    for fld, lvl in ((a, n), (b, n - 1)):
        glob = fld.gather_inner_data()
        if rank == 0:
            bad = int(np.count_nonzero(glob != history[lvl][1:NY + 1, 1:NX + 1]))
            if bad:
                print(f"ERROR rank 0: gather_inner_data of time level {lvl}: {bad} cells differ from the undivided field", flush=True)
                errors += bad
    # (3) the distributed shallow-water step (three fields per message, eight directions) against the oracle's step on
    #     the undivided domain; nine fields, leapfrog rotation, the joined and the time-loop entry alternately
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F, HG = {}, {}
    for k, nm in enumerate(names):
        F[nm] = D.r2d_field(g, pts[nm[0]])
        D.psy.hash_init(F[nm], SEED + 200 + k, box=ring)
        F[nm].data.mul_(0.01)
        F[nm].data.add_(1.0 if nm[0] == "p" else -0.005)
        hg = O.hash_field(SEED + 200 + k, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
        hg *= 0.01
        hg += 1.0 if nm[0] == "p" else -0.005
        HG[nm] = hg
    torch.cuda.synchronize()
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
    cur, old, new = names[:3], names[3:6], names[6:]
    for k in range(STEPS):
        fn = D.psy.invoke_shallow_step_dm_pipelined if k % 2 else D.psy.invoke_shallow_step_dm
        fn(prm, *[F[q] for q in cur + old + new], stream=s)
        O.sw_step(prm, gld, (2, NX + 1, 2, NY + 1), *[HG[q] for q in cur + old], *[HG[q] for q in new])
        cur, old, new = new, cur, old
    D.psy.halo_join(g, stream=s)
    s.synchronize()
    for q in cur:        # the newest level: internal region + the whole halo ring (corners included: nine-point footprint)
        got = F[q].get_data()[it.ystart - 2:it.ystop + 1, it.xstart - 2:it.xstop + 1]
        want = HG[q][gy0 + it.ystart - 2:gy0 + it.ystop + 1, gx0 + it.xstart - 2:gx0 + it.xstop + 1]
        bad = int(np.count_nonzero(got != want))
        if bad:
            print(f"ERROR rank {rank}: shallow-water step, field {q}: {bad} cells differ from the undivided oracle", flush=True)
            errors += bad
    # (4) halo_exchange_multi of SIX fields: more than a mailbox has room for (3), so the exchange goes in two turns
    six = [F[q] for q in names[:6]]
    for k, f in enumerate(six):
        D.psy.hash_init(f, SEED + 400 + k, box=ring)        # a wrong field everywhere ...
        D.psy.hash_init(f, SEED + 500 + k, box=it)          # ... the right one on the internal region
    D.psy.halo_exchange_multi(six)
    torch.cuda.synchronize()
    for k, f in enumerate(six):
        got = f.get_data()[it.ystart - 2:it.ystop + 1, it.xstart - 2:it.xstop + 1]
        right = O.hash_field(SEED + 500 + k, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
        want = O.hash_field(SEED + 400 + k, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
        want[1:NY + 1, 1:NX + 1] = right[1:NY + 1, 1:NX + 1]
        want = want[gy0 + it.ystart - 2:gy0 + it.ystop + 1, gx0 + it.xstart - 2:gx0 + it.xstop + 1]
        bad = int(np.count_nonzero(got != want))
        if bad:
            print(f"ERROR rank {rank}: halo_exchange_multi of six fields, field {k}: {bad} cells differ", flush=True)
            errors += bad
# ---- operations of ONE plan whose direction masks DIFFER ---------------------------------------------
# The reference's comm1..comm4 allow one-directional exchanges, and an edges-only Jacobi step shares its plan with full
# eight-direction exchanges.  Round `rnd`: operation N = a time-loop step whose JOIN comes late on one rank (the victim
# sleeps between step and join, so the strips of N sit unread in its mailbox half N & 1), operation N+1 = an exchange in ONE
# direction (three of the victim's four neighbours receive nothing from it), operation N+2 = an exchange of all eight,
# which stores into the half N & 1 again.  A neighbour that did not have to wait for the victim in N+1 would overwrite the
# strips of N before the victim has read them: every operation therefore raises and awaits the flags of ALL the plan's
# messages (peer_in_strips, dlesm_halo.hip).  Three different fields, so that a strip of the wrong operation shows.
import time  # noqa: E402
f2, f3 = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
edge_of_bit = {0: (it.xstop + 1, it.xstop + 1, it.ystart, it.ystop),      # Iplus: the east halo column
               1: (it.xstart - 1, it.xstart - 1, it.ystart, it.ystop),    # Iminus: the west one
               2: (it.xstart, it.xstop, it.ystop + 1, it.ystop + 1),      # Jplus: the north halo row
               3: (it.xstart, it.xstop, it.ystart - 1, it.ystart - 1)}    # Jminus: the south one


def halo_want(seed_wrong, seed_right, boxes):
    """the field as it must look (this rank's window incl. the halo ring): `wrong` everywhere, `right` on the internal
    region and on the cells of `boxes` (local 1-based inclusive) that lie inside the global interior"""
    right = O.hash_field(seed_right, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
    want = O.hash_field(seed_wrong, NY + 2, NX + 2, 0, 0, 1, NX + 2, 1, NY + 2)
    want[gy0 + it.ystart - 1:gy0 + it.ystop, gx0 + it.xstart - 1:gx0 + it.xstop] = \
        right[gy0 + it.ystart - 1:gy0 + it.ystop, gx0 + it.xstart - 1:gx0 + it.xstop]
    for (x0, x1, y0, y1) in boxes:
        ga, gb = max(gy0 + y0 - 1, 1), min(gy0 + y1 - 1, NY)            # rows / columns of the global (padded) array
        gc, gd = max(gx0 + x0 - 1, 1), min(gx0 + x1 - 1, NX)
        if ga <= gb and gc <= gd:
            want[ga:gb + 1, gc:gd + 1] = right[ga:gb + 1, gc:gd + 1]
    return want[gy0 + it.ystart - 2:gy0 + it.ystop + 1, gx0 + it.xstart - 2:gx0 + it.xstop + 1]


for rnd in range(4):
    victim, bit = rnd % world, rnd % 4
    for k, f in ((2, f2), (3, f3)):
        D.psy.hash_init(f, SEED + 600 + 10 * rnd + k, box=ring, stream=s)
        D.psy.hash_init(f, SEED + 700 + 10 * rnd + k, box=it, stream=s)
    D.psy.invoke_jacobi5_dm_pipelined(b, a, stream=s)               # operation N (edges only): its halos stay in the mailbox ...
    if rank == victim:
        s.synchronize()
        time.sleep(0.25)                                            # ... unread on the victim, while the others run ahead
    D.psy.halo_join(g, stream=s)
    f2.halo_exchange(1, stream=s, dirs=1 << bit)                    # operation N+1: one direction
    f3.halo_exchange(1, stream=s)                                   # operation N+2: all eight, the mailbox half of N again
    s.synchronize()
    a, b = b, a
    n += 1
    errors += check(a, n, f"time-loop step whose join came late on rank {victim} (masks vary, round {rnd})")
    win = (slice(it.ystart - 2, it.ystop + 1), slice(it.xstart - 2, it.xstop + 1))
    bad = int(np.count_nonzero(f2.get_data()[win] != halo_want(SEED + 600 + 10 * rnd + 2, SEED + 700 + 10 * rnd + 2, [edge_of_bit[bit]])))
    if bad:
        print(f"ERROR rank {rank}: one-direction exchange (bit {bit}), round {rnd}: {bad} cells differ", flush=True)
        errors += bad
    bad = int(np.count_nonzero(f3.get_data()[win] != halo_want(SEED + 600 + 10 * rnd + 3, SEED + 700 + 10 * rnd + 3, [ring.box()])))
    if bad:
        print(f"ERROR rank {rank}: eight-direction exchange behind it, round {rnd}: {bad} cells differ", flush=True)
        errors += bad
del f2, f3
if L.dlesm_ipc_open_retries():
    print(f"ERROR rank {rank}: hipIpcOpenMemHandle had to be retried (or a gather fell back to host memory) "
          f"{L.dlesm_ipc_open_retries()} time(s) although the importers take turns -- see the log above", flush=True)
    errors += 1
if L.dlesm_wait_timed_out(0):
    print(f"ERROR rank {rank}: a device-side wait gave up", flush=True)
    errors += 1
t = torch.tensor([errors])
dist.all_reduce(t)
dist.barrier()
print(f"rank {rank}: tile {it.nx}x{it.ny} of {NX}x{NY}, {n} steps, errors {errors} (all ranks {int(t.item())})", flush=True)
D.parallel_finalise()
dist.destroy_process_group()
sys.exit(1 if int(t.item()) else 0)
