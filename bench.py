#!/usr/bin/env python3
"""Headline benchmark: fp64 5-point Jacobi over a dl_esm_inf r2d_field (+ RCCL halo exchange
when N > 1), BASELINE.json metric "stencil Mcells/s + achieved HBM GB/s, 16384^2 fp64 grid".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tile 16384] [--alignment 64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: (halo exchange +) one Jacobi sweep of the
per-GPU tile.  Weak scaling: every GPU owns a tile x tile T-point field; the global domain is
(tile*P) x (tile*Q) with P x Q what go_decompose picks for N ranks.  Inputs are generated on
the device (counter-based hash of the global cell index) before the timed region starts.

Prints ONE JSON line (rank 0).  The CPU oracle is used only for the `cpu_baseline` leg.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_CELL = 16            # Jacobi-5: 8 B compulsory read + 8 B write
SEED = 20261004


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tile", type=int, default=16384, help="per-GPU interior is tile x tile")
    ap.add_argument("--alignment", type=int, default=64, help="DL_ESM_ALIGNMENT for the run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--fused", type=int, default=1,
                    help="1 GPU only: advance this many time steps per launch (2..8, temporal blocking); "
                         "the headline run keeps 1 = one sweep per time step")
    ap.add_argument("--no-temporal-blocking", action="store_true", help="skip the secondary fused-steps figure")
    ap.add_argument("--no-plan", action="store_true", help="skip the launch-shape planning call before the warm-up")
    ap.add_argument("--no-shallow", action="store_true", help="skip the secondary shallow-water figure")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE Jacobi configurations")
    ap.add_argument("--no-weak-tile", action="store_true",
                    help="skip the 8192^2-per-GPU weak-scaling object (BASELINE configs[4]'s tile)")
    ap.add_argument("--no-peer", action="store_true", help="skip the peer-transport object (N > 1: beside RCCL on the "
                    "weak-scaling tile; N = 1: 4096^2 loop-back)")
    ap.add_argument("--dm-form", choices=["timeloop", "joined", "safe"], default="timeloop",
                    help="N > 1: form of the distributed step the headline times.  timeloop (default): one launch per step, "
                         "the exchange of step k joined on the device by step k+1's frame workgroups behind ONE agent-scope "
                         "acquire (the consumer form of MI355X_MICROARCH.md), one host join closing the loop; joined: every "
                         "step joins its own exchange on the caller's stream (kernel boundaries only); safe: DLESM_DM_SAFE -- "
                         "frame in its own launch, events, unpack into the field.  (--tune dm_acquire=0 is the round-3 "
                         "time loop without the acquire: faster by a fraction of a percent, not guide-valid.)")
    ap.add_argument("--selfcheck-steps", type=int, default=50,
                    help="N > 1: back-to-back steps of the timed form compared with as many stencil + exchange steps before timing")
    ap.add_argument("--force-dm-leg", action="store_true",
                    help="rehearsal on 1 GPU: run the N>1 secondary leg with a 1-rank process group")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="experiments: dlesm_set_tuning(KEY, INT) before the run (repeatable)")
    ap.add_argument("--rows", type=int, default=None, help="tuning: rows per strip")
    ap.add_argument("--variant", type=int, default=None, help="tuning: kernel variant bits")
    return ap.parse_args()


def cpu_baseline(host_in, ld, box, budget_s):
    """time the ORACLE (oracle/dlesm_oracle.c, the GOcean-form CPU loops) on this box's host
    cores, on a bounded sample: whole-tile sweeps until the budget is used"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    xs, xe, ys, ye = box
    cells = (xe - xs + 1) * (ye - ys + 1)
    threads = min(O.host_threads(), int(os.environ.get("DLESM_CPU_THREADS", "16")))
    res = {}
    for label, nthr, share in (("all", threads, 0.6), ("one", 1, 0.4)):
        # fresh pages, first touched by the threads that will stream them
        src, out = np.empty_like(host_in), np.empty_like(host_in)
        O.lib().orc_copy_rows_omp(src, host_in, ld, host_in.shape[0], nthr)
        O.lib().orc_copy_rows_omp(out, host_in, ld, host_in.shape[0], nthr)
        host_in_t = src
        O.jacobi5(host_in_t, out, ld, xs, xe, ys, ye, threads=nthr)     # warm
        t0, sweeps = time.perf_counter(), 0
        while True:
            O.jacobi5(host_in_t, out, ld, xs, xe, ys, ye, threads=nthr)
            sweeps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s * share or sweeps >= 2000:
                break
        res[label] = (cells * sweeps / dt / 1e6, sweeps, dt)
    out_d = {
        "value": round(res["all"][0], 1), "unit": "Mcells/s", "cores": threads, "kind": "port",
        "single_core_value": round(res["one"][0], 1),
        "sample": f"oracle orc_jacobi5 (C, gcc -O3, GOcean kernel form) on the same "
                  f"{xe - xs + 1}x{ye - ys + 1} tile: {res['all'][1]} sweeps in {res['all'][2]:.1f}s "
                  f"with {threads} OpenMP threads, {res['one'][1]} sweeps in {res['one'][2]:.1f}s on 1 core",
    }
    # the same step the way a GOcean application runs it on the CPU: Fortran pointwise kernel called from
    # the PSy loop nest, OpenMP over jj (oracle/cpu_psy_loops.f90, amdflang -O3); a quarter of the budget
    try:
        src, out = np.empty_like(host_in), np.empty_like(host_in)
        O.lib().orc_copy_rows_omp(src, host_in, ld, host_in.shape[0], threads)
        O.lib().orc_copy_rows_omp(out, host_in, ld, host_in.shape[0], threads)
        O.jacobi5_fortran(src, out, ld, xs, xe, ys, ye, threads=threads)
        t0, sweeps = time.perf_counter(), 0
        while True:
            O.jacobi5_fortran(src, out, ld, xs, xe, ys, ye, threads=threads)
            sweeps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s * 0.25 or sweeps >= 2000:
                break
        out_d["fortran_psy_loops_value"] = round(cells * sweeps / dt / 1e6, 1)
        out_d["sample"] += f"; Fortran PSy loops (amdflang -O3, {threads} threads): {sweeps} sweeps in {dt:.1f}s"
    except Exception as e:                                # noqa: BLE001  (a missing Fortran runtime must not cost the line)
        out_d["fortran_psy_loops_value"] = None
        out_d["fortran_psy_loops_error"] = f"{type(e).__name__}: {e}"
    return out_d


XT_ROWS = {2: 4, 3: 6, 4: 8, 5: 8, 6: 12, 7: 16, 8: 16}      # rows per wave tile the library picks per T
TB_STEPS = 8                                                 # time steps per launch of the secondary legs
SW_KERNEL = "shallow_tile<2,dpp,nt>"                            # what dlesm_shallow_step_f64 launches by default
SW_DX, SW_DY, SW_DT = 1.0e5, 1.0e5, 90.0                      # frozen constants of the shallow-water legs
WEAK_TILE = 8192                                             # the per-GPU tile BASELINE configs[4] names
MIN_SECONDARY_LAUNCHES = 24                                  # every secondary leg times at least this many launches
TRAFFIC_SOURCE = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with the "
                  "guide's gfx950 correction (FETCH_SIZE x 2), per launch; read from the committed file, "
                  "NOT re-measured in this run")


def baseline_config(tile, world=1):
    """which entry of BASELINE.json `configs` a Jacobi tile corresponds to"""
    if world > 1:
        return "BASELINE configs[4] form" + ("" if tile == 8192 else f" with a {tile}^2 tile (configs[4] names 8192^2)")
    return {4096: "BASELINE configs[1]", 16384: "BASELINE configs[2]",
            8192: "the per-GPU tile of BASELINE configs[4], on one GPU"}.get(tile, "not a BASELINE size")


def traffic_for(tile, alignment, fused=1):
    """PMC-measured fabric bytes per launch for this configuration, if profiles/traffic.json has it"""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(tj))
        key = f"{tile}x{tile}/A{alignment}" + (f"/fused{fused}" if fused > 1 else "")
        return rec.get(key, {}).get("hbm_bytes_per_launch")
    except Exception:                                        # noqa: BLE001
        return None


def make_grid(D, nx, ny, alignment, halo_width=1):
    os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(nx, ny, halo_width=halo_width)
    D.grid_init(g, 1.0, 1.0)
    return g


def jacobi_config(D, torch, stream, tile, alignment, steps, warmup=10):
    """Secondary figure (never `value`): one more BASELINE Jacobi configuration on this GPU, timed
    like the headline (planning call, warm-up, HIP events on the launch stream)."""
    steps = max(steps, MIN_SECONDARY_LAUNCHES)
    g = make_grid(D, tile, tile, alignment)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        D.psy.hash_init(a, SEED, stream=stream)
        D.copy_field(a, b, stream=stream)
        D.psy.autotune_jacobi5(b, a, stream=stream)
        D.copy_field(a, b, stream=stream)
        for _ in range(warmup):
            D.psy.invoke_jacobi5(b, a, stream=stream)
            a, b = b, a
        # a timed region of at least ~40 ms whatever the tile (a 4096^2 step takes 45 us: 100 of them are over
        # before the clocks have settled): size it from a short estimate, at most 2000 launches
        e0.record(stream)
        for _ in range(20):
            D.psy.invoke_jacobi5(b, a, stream=stream)
            a, b = b, a
        e1.record(stream)
        stream.synchronize()
        est = e0.elapsed_time(e1) / 20
        steps = min(2000, max(steps, int(40.0 / max(est, 1e-3))))
        e0.record(stream)
        for _ in range(steps):
            D.psy.invoke_jacobi5(b, a, stream=stream)
            a, b = b, a
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / steps
    cells = tile * tile
    gbs = BYTES_PER_CELL * cells / (ms * 1e-3) / 1e9
    out = {"workload": f"jacobi5 {tile}x{tile} fp64, DL_ESM_ALIGNMENT={alignment} (ld {g.nx}; {baseline_config(tile)})",
           "tile": tile, "DL_ESM_ALIGNMENT": alignment, "ld": g.nx, "steps": steps,
           "value": round(cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s", "ms_per_step": round(ms, 5),
           # two ping-pong arrays below ~150 MB each largely live in the 256 MiB Infinity Cache (the library keeps cached
           # stores there for that reason): such a configuration is NOT bounded by HBM alone, and says so
           "roofline": {"bound": "hbm" if g.nx * g.ny * 8 >= (150 << 20) else "hbm+infinity-cache",
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic_for(tile, alignment),
                        "traffic_source": TRAFFIC_SOURCE, "algorithmic_bytes_per_launch": BYTES_PER_CELL * cells},
           "checksum": D.field_checksum(a)}
    del a, b
    torch.cuda.empty_cache()
    return out


def copy_ceiling(D, torch, stream, srcs, dsts, n_doubles, launches=MIN_SECONDARY_LAUNCHES):
    """What a linear sweep with THIS many concurrent HBM streams reaches on THIS box, in THIS process: nread arrays
    read once, nwrite arrays written once, nothing else (dlesm_lab_stream_copy_f64: one 16-byte element per thread per
    array, workgroups sweeping memory front to back), with default and with non-temporal stores.  The kernels'
    `frac_of_copy_ceiling` is measured against the better of the two.  srcs / dsts: device tensors (clobbered: dsts)."""
    L = D._cabi.lib()
    nr, nw = len(srcs), len(dsts)
    # 16-byte elements: an odd element count (the reference's DEFAULT alignment: ld 16387 x 16387 rows) is rounded down --
    # one double of 268 million does not move the rate, and an odd count made the leg fail in round 3
    n_doubles &= ~1
    sp = (C.c_void_p * nr)(*[t.data_ptr() for t in srcs])
    dp = (C.c_void_p * nw)(*[t.data_ptr() for t in dsts])
    out = {"sweep": f"{nr} arrays read + {nw} written, {n_doubles * 8 / 1e6:.0f} MB each, linear, 16 B per thread per array",
           "launches": launches}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for label, nt in (("default_stores", 0), ("nt_stores", 2)):
        with torch.cuda.stream(stream):
            for k in range(launches + 4):
                if k == 4:
                    e0.record(stream)
                D._cabi.check_lab(D._cabi.lab().dlesm_lab_stream_copy_f64(nr, nw, sp, dp, n_doubles, nt, C.c_void_p(stream.cuda_stream)))
            e1.record(stream)
        stream.synchronize()
        ms = e0.elapsed_time(e1) / launches
        gbs = 8.0 * (nr + nw) * n_doubles / (ms * 1e-3) / 1e9
        out[label] = {"ms": round(ms, 5), "gbs": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
        best = max(best, gbs)
    out["best_gbs"] = round(best, 1)
    return out


def temporal_blocking(D, torch, grid, a, stream, steps, tile, T=TB_STEPS):
    """Secondary figure (never `value`): the same time steps advanced T per sweep by the fused
    kernel (dlesm_stencil5_multi_f64).  First T single steps and one fused launch from the same
    state must agree bit for bit, then steps//T launches are timed with events on the stream."""
    x, y, z = (D.r2d_field(grid, D.GO_T_POINTS) for _ in range(3))
    with torch.cuda.stream(stream):
        for f in (x, y, z):
            D.copy_field(a, f, stream=stream)            # same fixed boundary ring everywhere
        src, dst = x, y
        for _ in range(T):
            D.psy.invoke_jacobi5(dst, src, stream=stream)
            src, dst = dst, src
        D.psy.invoke_jacobi5_multi(z, a, T, stream=stream)
    stream.synchronize()
    same = bool(torch.equal(src.data, z.data))
    # the clocks drop while the host compares the arrays: warm up again, and time enough launches
    launches = max(MIN_SECONDARY_LAUNCHES, steps // T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(10):
            D.psy.invoke_jacobi5_multi(y, x, T, stream=stream)
            x, y = y, x
        e0.record(stream)
        for _ in range(launches):
            D.psy.invoke_jacobi5_multi(y, x, T, stream=stream)
            x, y = y, x
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / launches
    cells = tile * tile
    return {"fused_steps": T, "value": round(cells * T / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s",
            "steps": launches * T, "ms_per_launch": round(ms, 5), "ms_per_step": round(ms / T, 5),
            "hbm_gbs": round(BYTES_PER_CELL * cells / (ms * 1e-3) / 1e9, 1),
            "algorithmic_bytes_per_launch": BYTES_PER_CELL * cells,
            "bit_identical_to_single_steps": same, "kernel": f"jacobi5xt_tile<{T},{XT_ROWS[T]},dpp>"}


def shallow_water(D, torch, stream, alignment, tile=8192, steps=40, cpu_seconds=0.0, plan=True):
    """Secondary figure (never `value`): BASELINE configs[3], the fused shallow-water u/v/h step
    (9-point composite footprint, 72 B/cell algorithmic) on a tile x tile C-grid, leapfrog rotation
    of the three time levels between steps.  With cpu_seconds > 0 its own cpu_baseline: the oracle's
    GOcean kernel sequence (orc_sw_step, 1 core) on a 256-row slab of the same initial state, whose
    result rows must equal the GPU's bit for bit."""
    steps = max(steps, MIN_SECONDARY_LAUNCHES)
    os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(tile, tile)
    D.grid_init(g, SW_DX, SW_DY)            # the kernels' GO_GRID_DX_CONST / DY_CONST arguments come from the grid
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    F = {}
    with torch.cuda.stream(stream):
        for k, name in enumerate(names):
            F[name] = D.r2d_field(g, pts[name[0]])
            D.psy.hash_init(F[name], SEED + k, stream=stream)
            F[name].data.add_(1.0 if name[0] == "p" else -0.5)     # p in [1,2), u, v in [-0.5,0.5)
    prm = D.psy.shallow_params(SW_DX, SW_DY, SW_DT)
    it = F["p"].internal
    cur, old, new = [F["u"], F["v"], F["p"]], [F["uold"], F["vold"], F["pold"]], [F["unew"], F["vnew"], F["pnew"]]
    # slab kept for the CPU leg: inputs rows j0-1 .. j0+h, first-step outputs rows j0 .. j0+h-1 (1-based j0)
    h = min(256, it.ny)
    j0 = it.ystart + (it.ny - h) // 2
    slab_in = slab_out = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        if cpu_seconds > 0:
            slab_in = [f.data[j0 - 2:j0 + h, :].clone() for f in cur + old]
        D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=stream)
        if cpu_seconds > 0:
            slab_out = [f.data[j0 - 1:j0 + h - 1, :].clone() for f in new]
        if plan:        # planning call (like the headline's): launch shape + cache policy, same bits whatever it picks
            D.psy.autotune_shallow(prm, *cur, *old, *new, stream=stream)
        old, cur, new = cur, new, old
        for _ in range(5):
            D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=stream)
            old, cur, new = cur, new, old
        e0.record(stream)
        for _ in range(steps):
            D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=stream)
            old, cur, new = cur, new, old
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / steps
    cells, bpc = tile * tile, 72
    gbs = bpc * cells / (ms * 1e-3) / 1e9
    out = {"workload": f"shallow-water u/v/h fused step {tile}x{tile} fp64 (BASELINE configs[3])", "steps": steps,
           "value": round(cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s", "ms_per_step": round(ms, 5),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": bpc,
                        "algorithmic_bytes_per_launch": bpc * cells, "kernel": SW_KERNEL},
           "checksum_pnew": D.field_checksum(cur[2])}
    if cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_lib as O
        hin = [t.cpu().numpy() for t in slab_in]
        want = [np.zeros_like(hin[0]) for _ in range(3)]
        box = (it.xstart, it.xstop, 2, h + 1)
        O.sw_step(prm, g.nx, box, *hin, *want)                      # warm + the check
        got = [t.cpu().numpy() for t in slab_out]
        same = all(np.array_equal(w[1:h + 1, it.xstart - 1:it.xstop], gg[:, it.xstart - 1:it.xstop])
                   for w, gg in zip(want, got))
        t0, n = time.perf_counter(), 0
        while True:
            O.sw_step(prm, g.nx, box, *hin, *want)
            n += 1
            dt = time.perf_counter() - t0
            if dt > cpu_seconds * 0.5 or n >= 500:
                break
        one = it.nx * h * n / dt / 1e6
        # ... and the way a GOcean application runs the step on the host: pointwise Fortran kernels called from the seven
        # PSy loop nests, OpenMP `parallel do` over jj (oracle/cpu_psy_loops.f90), all the cores this process may use
        threads = min(O.host_threads(), int(os.environ.get("DLESM_CPU_THREADS", "16")))
        fwant = [np.zeros_like(hin[0]) for _ in range(3)]
        scratch = [np.zeros_like(hin[0]) for _ in range(4)]
        try:
            O.sw_step_fortran(prm, g.nx, box, *hin, *fwant, threads=threads, scratch=scratch)     # warm, first touch
            fsame = all(np.array_equal(a[1:h + 1], b[1:h + 1]) for a, b in zip(fwant, want))
            t1, m = time.perf_counter(), 0
            while True:
                O.sw_step_fortran(prm, g.nx, box, *hin, *fwant, threads=threads, scratch=scratch)
                m += 1
                dtf = time.perf_counter() - t1
                if dtf > cpu_seconds * 0.5 or m >= 2000:
                    break
            allv, fnote = it.nx * h * m / dtf / 1e6, f"{m} steps in {dtf:.1f}s with {threads} OpenMP threads"
        except Exception as e:                               # noqa: BLE001  (a missing Fortran runtime must not cost the line)
            allv, fsame, threads, fnote = one, None, 1, f"Fortran PSy loops unavailable ({type(e).__name__}: {e}); 1-core C value"
        out["cpu_baseline"] = {"value": round(allv, 1), "unit": "Mcells/s", "cores": threads, "kind": "port",
                               "single_core_value": round(one, 1),
                               "sample": f"the un-fused GOcean kernel sequence (cu, cv, z, h, unew, vnew, pnew) on a {it.nx}x{h} "
                                         f"slab of the same initial state: Fortran pointwise kernels in seven PSy loop nests, "
                                         f"amdflang -O3, {fnote}; oracle orc_sw_step (C, gcc -O3): {n} steps in {dt:.1f}s on 1 core",
                               "fortran_loops_equal_c_oracle": fsame,
                               "gpu_first_step_equals_oracle_on_slab": bool(same)}
    # the ceiling of THIS stream count on this box, same arrays, same process: six arrays read + three written
    try:
        cc = copy_ceiling(D, torch, stream, [f.data for f in cur + old], [f.data for f in new], g.nx * g.ny)
        out["copy_ceiling"] = cc
        out["roofline"]["frac_of_copy_ceiling"] = round(gbs / cc["best_gbs"], 4)
    except Exception as e:                                   # noqa: BLE001
        out["copy_ceiling"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["unfused"] = shallow_water_unfused(D, torch, stream, g, F, prm, tile, steps, ms)
    except Exception as e:                                   # noqa: BLE001
        out["unfused"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["with_time_smooth"] = shallow_water_smooth(D, torch, stream, g, F, prm, tile, steps)
    except Exception as e:                                   # noqa: BLE001
        out["with_time_smooth"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["two_steps_per_launch"] = shallow_water_x2(D, torch, stream, g, F, prm, tile, steps, ms)
    except Exception as e:                                   # noqa: BLE001
        out["two_steps_per_launch"] = {"error": f"{type(e).__name__}: {e}"}
    del F, cur, old, new
    torch.cuda.empty_cache()
    try:
        out["sw_offset_periodic"] = shallow_water_periodic(D, torch, stream, alignment, tile, steps)
    except Exception as e:                                   # noqa: BLE001
        out["sw_offset_periodic"] = {"error": f"{type(e).__name__}: {e}"}
    return out


SW_KERNEL_BYTES = {"cu": 24, "cv": 24, "z": 32, "h": 32, "unew": 40, "vnew": 40, "pnew": 32}   # 8 B x (arrays read + 1)


def shallow_water_unfused(D, torch, stream, g, F, prm, tile, steps, fused_ms):
    """What an UNMODIFIED generated PSy layer gets: the same time step as seven launches, one per GOcean kernel
    (compute_cu, cv, z, h, unew, vnew, pnew -- dlesm_compute_*_f64), every intermediate through HBM: 224 B/cell
    algorithmic against 72 for the fused step.  The sequence must reproduce the fused step bit
    for bit from the same state; then each kernel is timed on its own and the sequence as a whole."""
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    tdt = SW_DT + SW_DT
    with torch.cuda.stream(stream):
        for k, n in enumerate(names[:6]):                            # a sane state again (the ceiling sweeps wrote sums)
            D.psy.hash_init(F[n], SEED + k, stream=stream)
            F[n].data.add_(1.0 if n[0] == "p" else -0.5)
        for n, pt in (("cu", D.GO_U_POINTS), ("cv", D.GO_V_POINTS), ("z", D.GO_F_POINTS), ("h", D.GO_T_POINTS)):
            F[n] = D.r2d_field(g, pt)
        chk = [D.r2d_field(g, F[n].defined_on) for n in ("u", "v", "p")]
        for f in chk + [F[n] for n in names[6:]]:
            D.set_field(f, 0.0, stream=stream)
        D.psy.invoke_shallow_step(prm, *[F[n] for n in names[:6]], *chk, stream=stream)
        D.psy.invoke_shallow_kernel_sequence(tdt, *[F[n] for n in names[:6]], F["cu"], F["cv"], F["z"], F["h"],
                                             *[F[n] for n in names[6:]], stream=stream)
    stream.synchronize()
    same = all(bool(torch.equal(F[n].data, c.data)) for n, c in zip(names[6:], chk))
    del chk
    torch.cuda.empty_cache()
    cells = tile * tile
    xs, xe, ys, ye = F["p"].internal.box()
    grown = {"cu": (xs - 1, xe, ys, ye + 1), "cv": (xs, xe + 1, ys - 1, ye), "z": (xs - 1, xe, ys - 1, ye),
             "h": (xs, xe + 1, ys, ye + 1)}
    P = D.psy
    calls = {
        "cu": lambda: P.invoke_compute_cu(F["cu"], F["p"], F["u"], grown["cu"], stream),
        "cv": lambda: P.invoke_compute_cv(F["cv"], F["p"], F["v"], grown["cv"], stream),
        "z": lambda: P.invoke_compute_z(F["z"], F["p"], F["u"], F["v"], grown["z"], stream),
        "h": lambda: P.invoke_compute_h(F["h"], F["p"], F["u"], F["v"], grown["h"], stream),
        "unew": lambda: P.invoke_compute_unew(F["unew"], F["uold"], F["z"], F["cv"], F["h"], tdt, None, stream),
        "vnew": lambda: P.invoke_compute_vnew(F["vnew"], F["vold"], F["z"], F["cu"], F["h"], tdt, None, stream),
        "pnew": lambda: P.invoke_compute_pnew(F["pnew"], F["pold"], F["cu"], F["cv"], tdt, None, stream),
    }
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per = {}
    reps = max(10, steps // 2)
    for name, fn in calls.items():
        with torch.cuda.stream(stream):
            for k in range(reps + 3):
                if k == 3:
                    e0.record(stream)
                fn()
            e1.record(stream)
        stream.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = SW_KERNEL_BYTES[name] * cells / (ms * 1e-3) / 1e9
        per[name] = {"ms": round(ms, 5), "bytes_per_cell": SW_KERNEL_BYTES[name], "gbs": round(gbs, 1),
                     "frac": round(gbs / HBM_PEAK_GBS, 4)}
    with torch.cuda.stream(stream):
        for k in range(steps + 2):
            if k == 2:
                e0.record(stream)
            P.invoke_shallow_kernel_sequence(tdt, *[F[n] for n in names[:6]], F["cu"], F["cv"], F["z"], F["h"],
                                             *[F[n] for n in names[6:]], stream=stream)
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gbs = 224 * cells / (ms * 1e-3) / 1e9
    # time_smooth (the Asselin filter of the GOcean leapfrog, one launch per prognostic field): 3 read + 1 written
    with torch.cuda.stream(stream):
        for k in range(reps + 3):
            if k == 3:
                e0.record(stream)
            P.invoke_time_smooth(F["u"], F["unew"], F["uold"], 0.001, None, stream)
        e1.record(stream)
    stream.synchronize()
    ts = e0.elapsed_time(e1) / reps
    # each kernel beside the linear sweep of ITS stream count, same arrays, same process (2, 3 or 4 arrays read + 1 written):
    # the last thing this leg does with the intermediates (the sweeps write sums into cu)
    ceilings = {}
    try:
        srcs = [F[n].data for n in ("p", "u", "v", "h")]
        for nread in (2, 3, 4):
            cc = copy_ceiling(D, torch, stream, srcs[:nread], [F["cu"].data], g.nx * g.ny)
            ceilings[nread] = cc["best_gbs"]
        for name, rec in per.items():
            nread = SW_KERNEL_BYTES[name] // 8 - 1
            rec["copy_ceiling_gbs"] = ceilings[nread]
            rec["frac_of_copy_ceiling"] = round(rec["gbs"] / ceilings[nread], 4)
    except Exception as e:                                   # noqa: BLE001  (a diagnostic must not cost the leg)
        ceilings = {"error": f"{type(e).__name__}: {e}"}
    return {"workload": f"the same step as SEVEN launches, one per GOcean kernel (what an unmodified generated PSy layer "
                        f"runs), {tile}x{tile} fp64", "steps": steps,
            "value": round(cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s", "ms_per_step": round(ms, 5),
            "bit_identical_to_fused_step": same,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": 224,
                         "algorithmic_bytes_per_launch_sequence": 224 * cells,
                         "kernel": "swk_tile<compute_{cu,cv,z,h,unew,vnew,pnew}>"},
            "per_kernel": per,
            "per_kernel_copy_ceilings_gbs": {f"{k}_read_1_written": v for k, v in ceilings.items()} if "error" not in ceilings else ceilings,
            "time_smooth": {"ms": round(ts, 5), "bytes_per_cell": 32, "gbs": round(32 * cells / (ts * 1e-3) / 1e9, 1),
                            "frac": round(32 * cells / (ts * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            **({"frac_of_copy_ceiling": round(32 * cells / (ts * 1e-3) / 1e9 / ceilings[3], 4)} if 3 in ceilings else {})},
            "fusion_speedup": round(ms / fused_ms, 3)}


def shallow_water_x2(D, torch, stream, g, F, prm, tile, steps, fused_ms):
    """Secondary object (never `value`): TWO leapfrog steps per launch (dlesm_shallow_step_x2_f64) -- six arrays
    read and six written per two steps, 48 B/cell/step instead of 72 -- the one lever above the nine-stream ceiling the
    fused step sits on.  First both new levels against two single steps on the full tile (bit for bit), then >= 24
    launches with the four-level rotation, beside the 6-read + 6-written copy of the same arrays."""
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    A = [F[n] for n in names]
    for f in A[6:]:                                          # (the time-smooth leg before may have left anything here: a ring
        D.copy_field(A[0], f, stream=stream)                 #  that holds finite values is all the comparison needs)
    extra = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(6)]                  # level n+2, and three levels of the reference run
    n2, ref = extra[:3], extra[3:]
    with torch.cuda.stream(stream):
        for f in extra:
            D.copy_field(A[0], f, stream=stream)
        ref1 = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(3)]
        for f in ref1:
            D.copy_field(A[0], f, stream=stream)
        D.psy.invoke_shallow_step_x2(prm, *A, *n2, stream=stream)
        D.psy.invoke_shallow_step(prm, *A[:6], *ref1, stream=stream)
        D.psy.invoke_shallow_step(prm, *ref1, *A[:3], *ref, stream=stream)
    stream.synchronize()
    same = all(torch.equal(a.data, b.data) for a, b in zip(A[6:] + n2, ref1 + ref))
    del ref1, ref
    launches = max(MIN_SECONDARY_LAUNCHES, steps // 2)
    cur, old, l1, l2 = A[:3], A[3:6], A[6:], n2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for k in range(launches + 4):
            if k == 4:
                e0.record(stream)
            D.psy.invoke_shallow_step_x2(prm, *cur, *old, *l1, *l2, stream=stream)
            cur, old, l1, l2 = l2, l1, old, cur
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / launches
    cells = tile * tile
    gbs = 96 * cells / (ms * 1e-3) / 1e9
    out = {"what": "two leapfrog steps per launch: levels n+1 and n+2 from n and n-1, six arrays read + six written",
           "launches": launches, "time_steps": 2 * launches, "value": round(2 * cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s",
           "ms_per_launch": round(ms, 5), "ms_per_step": round(ms / 2, 5),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_cell_per_launch": 96, "algorithmic_bytes_per_cell_per_step": 48,
                        "kernel": "shallow_tile_x2<4,2> (four-row tiles, the four waves of a workgroup vertically adjacent)"},
           "bit_identical_to_two_single_steps": bool(same), "speedup_per_step_vs_fused_step": round(2 * fused_ms / ms, 3)}
    try:
        cc = copy_ceiling(D, torch, stream, [f.data for f in cur + old], [f.data for f in l1 + l2], g.nx * g.ny)
        out["copy_ceiling"] = cc
        out["roofline"]["frac_of_copy_ceiling"] = round(gbs / cc["best_gbs"], 4)
    except Exception as e:                                   # noqa: BLE001
        out["copy_ceiling"] = {"error": f"{type(e).__name__}: {e}"}
    del extra, n2
    torch.cuda.empty_cache()
    return out


def shallow_water_smooth(D, torch, stream, g, F, prm, tile, steps, alpha=0.001):
    """A WHOLE time step of the GOcean leapfrog -- the u/v/h update plus the Asselin filter (time_smooth) of the three old
    fields -- as ONE launch (dlesm_shallow_step_smooth_f64: 6 arrays read + 6 written = 96 B/cell) against the fused step
    followed by three time_smooth launches (72 + 3 x 32 = 168 B/cell).  Same bits, checked first; the benchmark's rotation
    (u <- unew; uold keeps the filtered u) between steps."""
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    cells = tile * tile
    it_box = F["p"].internal.box()
    with torch.cuda.stream(stream):
        for k, n in enumerate(names[:6]):
            D.psy.hash_init(F[n], SEED + k, stream=stream)
            F[n].data.add_(1.0 if n[0] == "p" else -0.5)
        chk = {n: D.r2d_field(g, F[n].defined_on) for n in names[3:]}
        for n in names[3:6]:
            D.copy_field(F[n], chk[n], stream=stream)
        cur = [F[n] for n in names[:3]]
        D.psy.invoke_shallow_step_smooth(prm, alpha, *cur, *[F[n] for n in names[3:]], stream=stream)
        D.psy.invoke_shallow_step(prm, *cur, *[chk[n] for n in names[3:]], stream=stream)
        for k in range(3):
            D.psy.invoke_time_smooth(cur[k], chk[names[6 + k]], chk[names[3 + k]], alpha, None, stream)
    stream.synchronize()
    same = all(bool(torch.equal(F[n].data, chk[n].data)) for n in names[3:])
    del chk
    torch.cuda.empty_cache()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for label in ("one_launch", "step_plus_three_time_smooth"):
        cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
        with torch.cuda.stream(stream):
            for k in range(steps + 3):
                if k == 3:
                    e0.record(stream)
                if label == "one_launch":
                    D.psy.invoke_shallow_step_smooth(prm, alpha, *cur, *old, *new, stream=stream)
                else:
                    D.psy.invoke_shallow_step(prm, *cur, *old, *new, stream=stream)
                    for j in range(3):
                        D.psy.invoke_time_smooth(cur[j], new[j], old[j], alpha, None, stream)
                cur, new = new, cur
            e1.record(stream)
        stream.synchronize()
        res[label] = e0.elapsed_time(e1) / steps
    # round 4: TWO whole filtered time steps per launch (dlesm_shallow_step_smooth_x2_f64: level n+2 and the filtered level n+1
    # from level n and the filtered level n-1; six arrays read + six written per two steps = 48 B/cell/step); checked against
    # two one-launch filtered steps from the same state with every array carrying the same boundary ring, then >= 24 launches
    x2 = None
    try:
        ex = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(3)]
        A = [F[n] for n in names]
        with torch.cuda.stream(stream):
            for k, n in enumerate(names[:6]):
                D.psy.hash_init(F[n], SEED + k, stream=stream)
                F[n].data.add_(1.0 if n[0] == "p" else -0.5)
            for k in range(3):                               # one ring for every level: u's, v's, p's
                keep = A[3 + k].data[it_box[2] - 1:it_box[3], it_box[0] - 1:it_box[1]].clone()
                D.copy_field(A[k], A[3 + k], stream=stream)
                A[3 + k].data[it_box[2] - 1:it_box[3], it_box[0] - 1:it_box[1]] = keep
                D.copy_field(A[k], A[6 + k], stream=stream)
                D.copy_field(A[k], ex[k], stream=stream)
            ref = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(9)]
            for k in range(9):
                D.copy_field(A[k], ref[k], stream=stream)
            D.psy.invoke_shallow_step_smooth_x2(prm, alpha, *A[:6], *A[6:], *ex, stream=stream)      # n+2 -> A[6:9], filtered n+1 -> ex
            rc, ro, rn = ref[:3], ref[3:6], ref[6:]
            for _ in range(2):
                D.psy.invoke_shallow_step_smooth(prm, alpha, *rc, *ro, *rn, stream=stream)
                rc, rn = rn, rc
        stream.synchronize()
        cut = lambda f: f.data[it_box[2] - 1:it_box[3], it_box[0] - 1:it_box[1]]      # noqa: E731
        same2 = all(bool(torch.equal(cut(a), cut(b))) for a, b in zip(A[6:] + ex, rc + ro))
        del ref, rc, ro, rn
        torch.cuda.empty_cache()
        launches = max(MIN_SECONDARY_LAUNCHES, steps // 2)
        cur, old, n2, o2 = A[:3], A[3:6], A[6:], ex
        with torch.cuda.stream(stream):
            for k in range(launches + 4):
                if k == 4:
                    e0.record(stream)
                D.psy.invoke_shallow_step_smooth_x2(prm, alpha, *cur, *old, *n2, *o2, stream=stream)
                cur, old, n2, o2 = n2, o2, cur, old
            e1.record(stream)
        stream.synchronize()
        ms2 = e0.elapsed_time(e1) / launches
        g2 = 96 * cells / (ms2 * 1e-3) / 1e9
        x2 = {"what": "two whole filtered time steps per launch: level n+2 and the filtered level n+1 from level n and the filtered level n-1",
              "launches": launches, "time_steps": 2 * launches, "value": round(2 * cells / (ms2 * 1e-3) / 1e6, 1), "unit": "Mcells/s",
              "ms_per_launch": round(ms2, 5), "ms_per_step": round(ms2 / 2, 5),
              "roofline": {"bound": "hbm", "achieved": round(g2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g2 / HBM_PEAK_GBS, 4),
                           "algorithmic_bytes_per_cell_per_launch": 96, "algorithmic_bytes_per_cell_per_step": 48,
                           "kernel": "shallow_tile_x2<3,2,smooth> (three-row tiles held to two waves per SIMD, four vertically adjacent tiles per workgroup)"},
              "bit_identical_to_two_one_launch_filtered_steps": bool(same2),
              "speedup_per_step_vs_one_launch_filtered_step": round(2 * res["one_launch"] / ms2, 3),
              "speedup_per_step_vs_step_plus_three_time_smooth": round(2 * res["step_plus_three_time_smooth"] / ms2, 3)}
        del ex
    except Exception as e:                                   # noqa: BLE001
        x2 = {"error": f"{type(e).__name__}: {e}"}
    ms = res["one_launch"]
    gbs = 96 * cells / (ms * 1e-3) / 1e9
    # the ceiling of THIS stream count in the same process: six arrays read, six written, three of them in place (the old
    # level), as the filtered step does -- the last thing this leg does with the arrays (the sweep clobbers them)
    try:
        cur, old, new = [F[n].data for n in names[:3]], [F[n].data for n in names[3:6]], [F[n].data for n in names[6:]]
        cc = copy_ceiling(D, torch, stream, cur + old, new + old, g.nx * g.ny)
        cc["sweep"] += " (three of the written arrays are read arrays: in place, as the filter of the old level)"
        cc_frac = round(gbs / cc["best_gbs"], 4)
    except Exception as e:      # noqa: BLE001  (a diagnostic must not cost the leg)
        cc, cc_frac = {"error": f"{type(e).__name__}: {e}"}, None
    return {"copy_ceiling": cc, "frac_of_copy_ceiling": cc_frac, "two_steps_per_launch": x2,
            "workload": f"one whole leapfrog step incl. the Asselin filter (time_smooth) of the old level, {tile}x{tile} fp64, "
                        "one launch per time step", "steps": steps,
            "value": round(cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s", "ms_per_step": round(ms, 5),
            "bit_identical_to_step_plus_time_smooth": same,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": 96,
                         "kernel": "shallow_tile<2,dpp,nt> with the filter folded in (6 arrays read, 6 written)"},
            "step_plus_three_time_smooth": {"ms_per_step": round(res["step_plus_three_time_smooth"], 5),
                                            "value": round(cells / (res["step_plus_three_time_smooth"] * 1e-3) / 1e6, 1),
                                            "algorithmic_bytes_per_cell": 168},
            "speedup": round(res["step_plus_three_time_smooth"] / ms, 3)}


def shallow_water_periodic(D, torch, stream, alignment, tile, steps):
    """The same update in the configuration of the GOcean `shallow` benchmark: SW offset, periodic in
    x and y (serial only in the reference).  A step = ONE launch, dlesm_shallow_step_sw_periodic_f64: the step
    over the internal region whose edge tiles also write the periodic images of the three new fields (round 2:
    the step + two copy launches), + leapfrog rotation."""
    os.environ["DL_ESM_ALIGNMENT"] = str(alignment)
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_PERIODIC, D.GO_BC_PERIODIC, D.GO_BC_NONE), D.GO_OFFSET_SW)
    g.decompose(tile, tile)
    D.grid_init(g, 1.0, 1.0)
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    F = {}
    with torch.cuda.stream(stream):
        for k, name in enumerate(["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]):
            F[name] = D.r2d_field(g, pts[name[0]])
            D.psy.hash_init(F[name], SEED + k, box=F[name].internal, stream=stream)
            F[name].data.add_(1.0 if name[0] == "p" else -0.5)
            D.psy.apply_periodic_halos(F[name], stream=stream)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    cur, old, new = [F["u"], F["v"], F["p"]], [F["uold"], F["vold"], F["pold"]], [F["unew"], F["vnew"], F["pnew"]]
    # leapfrog: S(p) of the even and of the odd time levels are conserved separately
    stream.synchronize()              # (the sums below run on torch's current stream, not on `stream`)
    p0 = [float(F[n].data[1:tile + 1, 1:tile + 1].sum().item()) for n in ("p", "pold")]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step():
        nonlocal cur, old, new
        D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *new, stream=stream)
        old, cur, new = cur, new, old

    # first: one step in the three-launch form must leave the same fields AND halos, bit for bit
    chk = [D.r2d_field(g, f.defined_on) for f in new]
    with torch.cuda.stream(stream):
        D.psy.invoke_shallow_step_sw(prm, *cur, *old, *chk, stream=stream)
        D.psy.apply_periodic_halos_multi(chk, stream=stream)
        D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *new, stream=stream)
    stream.synchronize()
    same = all(bool(torch.equal(a.data[:tile + 2, :tile + 2], b.data[:tile + 2, :tile + 2])) for a, b in zip(new, chk))
    del chk
    torch.cuda.empty_cache()
    with torch.cuda.stream(stream):
        D.psy.autotune_shallow_sw(prm, *cur, *old, *new, stream=stream)      # planning call, as for the NE step
        D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *new, stream=stream)   # `new` with its halos again
        for _ in range(5):
            step()
        e0.record(stream)
        for _ in range(steps):
            step()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / steps
    cells, bpc = tile * tile, 72
    gbs = bpc * cells / (ms * 1e-3) / 1e9
    p1 = float(cur[2].data[1:tile + 1, 1:tile + 1].sum().item())
    out = {"workload": f"shallow-water u/v/h step + periodic halo copies, {tile}x{tile} fp64, SW offset, periodic in x and y "
                       "(the GOcean `shallow` configuration)", "steps": steps,
           "value": round(cells / (ms * 1e-3) / 1e6, 1), "unit": "Mcells/s", "ms_per_step": round(ms, 5),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": bpc,
                        "kernel": "shallow_tile_sw<2,dpp,nt> with the periodic images stored by the edge tiles (one launch)"},
           "one_launch_equals_step_plus_halo_copies": same,
           # pnew - pold is a discrete divergence: on the torus SUM(p) is conserved up to rounding
           "mass_drift_relative": abs(p1 - p0[(5 + steps) % 2]) / abs(p0[(5 + steps) % 2])}
    # round 4: two plain steps per launch (dlesm_shallow_step_sw_x2_periodic_f64, 48 B/cell/step), checked against two one-launch steps
    try:
        ex = [D.r2d_field(g, f.defined_on) for f in cur + cur]          # levels n+1 and n+2 of the two-step call
        ref = [D.r2d_field(g, f.defined_on) for f in cur + cur]
        with torch.cuda.stream(stream):
            D.psy.invoke_shallow_step_sw_x2_periodic(prm, *cur, *old, *ex, stream=stream)
            D.psy.invoke_shallow_step_sw_periodic(prm, *cur, *old, *ref[:3], stream=stream)
            D.psy.invoke_shallow_step_sw_periodic(prm, *ref[:3], *cur, *ref[3:], stream=stream)
        stream.synchronize()
        same2 = all(bool(torch.equal(a.data[:tile + 2, :tile + 2], b.data[:tile + 2, :tile + 2])) for a, b in zip(ex, ref))
        del ref
        torch.cuda.empty_cache()
        launches = max(MIN_SECONDARY_LAUNCHES, steps // 2)
        with torch.cuda.stream(stream):
            c2, o2, n1, n2 = cur, old, ex[:3], ex[3:]
            for k in range(launches + 3):
                if k == 3:
                    e0.record(stream)
                D.psy.invoke_shallow_step_sw_x2_periodic(prm, *c2, *o2, *n1, *n2, stream=stream)
                c2, o2, n1, n2 = n2, n1, o2, c2
            e1.record(stream)
        stream.synchronize()
        ms2 = e0.elapsed_time(e1) / launches
        out["two_steps_per_launch"] = {"ms_per_launch": round(ms2, 5), "ms_per_step": round(ms2 / 2, 5),
                                       "value": round(2 * cells / (ms2 * 1e-3) / 1e6, 1), "unit": "Mcells/s",
                                       "algorithmic_bytes_per_cell_per_step": 48,
                                       "frac": round(96 * cells / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "kernel": "shallow_tile_sw_x2<2,2,plain> held to two waves per SIMD",
                                       "bit_identical_to_two_one_launch_steps_fields_and_halos": bool(same2),
                                       "speedup_per_step": round(2 * ms / ms2, 3)}
        del ex
    except Exception as e:                                   # noqa: BLE001
        out["two_steps_per_launch"] = {"error": f"{type(e).__name__}: {e}"}
    # round 4: the benchmark's WHOLE time loop -- update, Asselin filter of the old level, periodic images -- one launch per step
    # (dlesm_shallow_step_sw_smooth_periodic_f64, 96 B/cell/step) and TWO steps per launch (..._smooth_x2_periodic_f64, 48): the
    # two-step form is checked against two one-launch steps from the same state (fields and halos), then both are timed
    try:
        alpha = 0.001
        ex = [D.r2d_field(g, f.defined_on) for f in cur + cur]          # the second sextet of the ping-pong
        ref = [D.r2d_field(g, f.defined_on) for f in cur + old + new]
        with torch.cuda.stream(stream):
            for a, b in zip(cur + old + new, ref):
                D.copy_field(a, b, stream=stream)
            D.psy.invoke_shallow_step_sw_smooth_x2_periodic(prm, alpha, *cur, *old, *ex, stream=stream)
            rc, ro, rn = ref[:3], ref[3:6], ref[6:]
            for _ in range(2):
                D.psy.invoke_shallow_step_sw_smooth_periodic(prm, alpha, *rc, *ro, *rn, stream=stream)
                rc, rn = rn, rc
        stream.synchronize()
        cut = lambda f: f.data[:tile + 2, :tile + 2]      # noqa: E731
        same2 = all(bool(torch.equal(cut(a), cut(b))) for a, b in zip(ex, rc + ro))
        del ref, rc, ro, rn
        torch.cuda.empty_cache()
        res = {}
        with torch.cuda.stream(stream):
            c1, o1, n1 = cur, old, new
            for k in range(steps + 3):
                if k == 3:
                    e0.record(stream)
                D.psy.invoke_shallow_step_sw_smooth_periodic(prm, alpha, *c1, *o1, *n1, stream=stream)
                c1, n1 = n1, c1
            e1.record(stream)
        stream.synchronize()
        res["one"] = e0.elapsed_time(e1) / steps
        launches = max(MIN_SECONDARY_LAUNCHES, steps // 2)
        with torch.cuda.stream(stream):
            c2, o2, n2, q2 = cur, old, ex[:3], ex[3:]
            for k in range(launches + 3):
                if k == 3:
                    e0.record(stream)
                D.psy.invoke_shallow_step_sw_smooth_x2_periodic(prm, alpha, *c2, *o2, *n2, *q2, stream=stream)
                c2, o2, n2, q2 = n2, q2, c2, o2
            e1.record(stream)
        stream.synchronize()
        res["two"] = e0.elapsed_time(e1) / launches
        out["with_time_smooth"] = {
            "what": "the GOcean `shallow` benchmark's whole time step (update + time_smooth of the old level + periodic images)",
            "one_launch_per_step": {"ms_per_step": round(res["one"], 5), "value": round(cells / (res["one"] * 1e-3) / 1e6, 1),
                                    "algorithmic_bytes_per_cell_per_step": 96, "frac": round(96 * cells / (res["one"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "two_steps_per_launch": {"ms_per_launch": round(res["two"], 5), "ms_per_step": round(res["two"] / 2, 5),
                                     "value": round(2 * cells / (res["two"] * 1e-3) / 1e6, 1), "unit": "Mcells/s",
                                     "algorithmic_bytes_per_cell_per_step": 48,
                                     "frac": round(96 * cells / (res["two"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "kernel": "shallow_tile_sw_x2<3,2,smooth> (three-row tiles held to two waves per SIMD)",
                                     "bit_identical_to_two_one_launch_steps_fields_and_halos": bool(same2),
                                     "speedup_per_step": round(2 * res["one"] / res["two"], 3)}}
        del ex
    except Exception as e:                                   # noqa: BLE001
        out["with_time_smooth"] = {"error": f"{type(e).__name__}: {e}"}
    del F, cur, old, new
    torch.cuda.empty_cache()
    return out


def temporal_blocking_dm(D, torch, dist, tile, P, Q, stream, steps, T=TB_STEPS):
    """Secondary figure at N > 1: T time steps per call with ONE depth-T halo exchange
    (dlesm_jacobi5_multi_step_dm) on a decomposition made with halo_width = T.  Checked first,
    on every rank, against T x (single step + depth-T exchange) from the same state."""
    world = P * Q
    g = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    g.decompose(tile * P, tile * Q, halo_width=T)
    D.grid_init(g, 1.0, 1.0)
    x, y, u, v = (D.r2d_field(g, D.GO_T_POINTS) for _ in range(4))
    with torch.cuda.stream(stream):
        D.psy.hash_init(x, SEED)
        x.halo_exchange(1, stream=stream)                # depth T: the tables of this grid are depth-T
        for f in (y, u, v):
            D.copy_field(x, f, stream=stream)
        for _ in range(T):
            D.psy.invoke_jacobi5(v, u, stream=stream)
            v.halo_exchange(1, stream=stream)
            u, v = v, u
        D.psy.invoke_jacobi5_multi_dm(y, x, T, stream=stream)
    stream.synchronize()
    ok = torch.tensor([1 if torch.equal(u.data, y.data) else 0], device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    launches = max(MIN_SECONDARY_LAUNCHES, steps // T)
    with torch.cuda.stream(stream):
        for _ in range(10):
            D.psy.invoke_jacobi5_multi_dm(y, x, T, stream=stream)
            x, y = y, x
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(launches):
            D.psy.invoke_jacobi5_multi_dm(y, x, T, stream=stream)
            x, y = y, x
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wall = float(tt[0])
    cells = tile * tile * world
    return {"fused_steps": T, "value": round(cells * T * launches / wall / 1e6, 1), "unit": "Mcells/s",
            "steps": launches * T, "ms_per_launch": round(wall / launches * 1e3, 5),
            "ms_per_step": round(wall / launches / T * 1e3, 5), "halo_depth": T,
            "exchanges_per_step": round(1.0 / T, 3),
            "bit_identical_to_single_steps_plus_exchange": bool(int(ok[0])),
            "kernel": f"jacobi5xt_tile<{T},{XT_ROWS[T]},dpp> + one depth-{T} RCCL exchange per launch"}


def weak_scaling_tile(D, torch, dist, tile, alignment, world, P, Q, stream, steps, warmup=10, dm_form="timeloop"):
    """Secondary object on EVERY line (N = 1, 2, 4, 8): the Jacobi step on the per-GPU tile BASELINE
    configs[4] names (8192^2), global domain (tile*P) x (tile*Q) cut by go_decompose, RCCL halo exchange
    hidden behind the interior when N > 1 -- so that t(1)/t(N) for THAT tile can be read off the driver's
    own lines.  Timed like the headline: barrier + synchronize on both sides, max over ranks."""
    steps = max(steps, MIN_SECONDARY_LAUNCHES)
    g = make_grid(D, tile * P, tile * Q, alignment)
    a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
    it = a.internal
    assert (it.nx, it.ny) == (tile, tile) and (g.decomp.nx, g.decomp.ny) == (P, Q)
    # N > 1: the time-loop form of the distributed step -- the exchange of step k is joined by step k+1's
    # frame workgroups on the device, the caller's stream carries one launch per step; ONE join closes the loop
    dm_step = D.psy.invoke_jacobi5_dm_pipelined if dm_form == "timeloop" else D.psy.invoke_jacobi5_dm
    step = dm_step if world > 1 else D.psy.invoke_jacobi5
    L = D._cabi.lib()
    with torch.cuda.stream(stream):
        D.psy.hash_init(a, SEED, stream=stream)
        D.copy_field(a, b, stream=stream)
        a.halo_exchange(1, stream=stream)
        if world == 1:
            D.psy.autotune_jacobi5(b, a, stream=stream)
        else:
            D._cabi.check(L.dlesm_stencil5_autotune_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, it.xstart + 1,
                                                        it.xstop - 1, it.ystart + 1, it.ystop - 1,
                                                        C.c_void_p(stream.cuda_stream)))
        D.copy_field(a, b, stream=stream)
    stream.synchronize()
    same = None
    if world > 1:       # overlapped step == stencil + edge exchange, on every rank, before timing
        x, y = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
        with torch.cuda.stream(stream):
            D.copy_field(a, x, stream=stream)
            D.copy_field(a, y, stream=stream)
            D.psy.invoke_jacobi5(y, x, stream=stream)
            y.halo_exchange(1, stream=stream, dirs=D._cabi.DIRS_EDGES_ONLY)
            dm_step(b, a, stream=stream)
            D.psy.halo_join(g, stream=stream)
        stream.synchronize()
        ok = torch.tensor([1 if torch.equal(b.data, y.data) else 0], device="cuda")
        if os.environ.get("DLESM_BENCH_DEBUG") and not int(ok[0]):
            dd = (b.data != y.data).nonzero()
            print(f"[bench debug] rank {dist.get_rank()}: weak tile self-check: {dd.shape[0]} cells differ, first {dd[:8].tolist()} last "
                  f"{dd[-4:].tolist()}; internal ({it.xstart}:{it.xstop},{it.ystart}:{it.ystop})", file=sys.stderr, flush=True)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        same = bool(int(ok[0]))
        with torch.cuda.stream(stream):
            D.copy_field(a, b, stream=stream)
        del x, y

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            step(b, a, stream=stream)
            a, b = b, a
        if world > 1:
            D.psy.halo_join(g, stream=stream)
    # a secondary object: two timed windows of `steps` steps, the faster one reported (one 17-ms window of a run once read 22 ms
    # for no reason the kernels know; the headline keeps the contract's single window)
    walls = []
    for _ in range(2):
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(steps):
                step(b, a, stream=stream)
                a, b = b, a
            if world > 1:
                D.psy.halo_join(g, stream=stream)            # the one join of the loop, inside the timed region
        barrier()
        wall = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall = float(tt[0])
        walls.append(wall)
    wall = min(walls)
    cells = tile * tile * world
    ms = wall / steps * 1e3
    gbs = BYTES_PER_CELL * tile * tile / (ms * 1e-3) / 1e9
    out = {"workload": f"jacobi5 {tile}x{tile} fp64 T-field per GPU, {P}x{Q} decomposition "
                       f"({baseline_config(tile, world) if world > 1 else baseline_config(tile)})",
           "tile": tile, "n_gpus": world, "decomposition": f"{P}x{Q}", "global": [tile * P, tile * Q],
           "DL_ESM_ALIGNMENT": alignment, "steps": steps, "value": round(cells * steps / wall / 1e6, 1),
           "unit": "Mcells/s", "ms_per_step": round(ms, 5), "hbm_gbs_per_gpu": round(gbs, 1),
           "frac_of_hbm_peak_per_gpu": round(gbs / HBM_PEAK_GBS, 4), "scaling": "weak",
           "halo_exchange": f"rccl send/recv of the four edges, overlapped ({dm_form} form)" if world > 1 else "none (1 tile)",
           "timed_windows_s": [round(w, 6) for w in walls],
           "dm_step_equals_stencil_plus_exchange": same, "checksum": D.field_checksum(a)}
    del a, b
    torch.cuda.empty_cache()
    return out


def shallow_water_dm(D, torch, dist, tile, alignment, world, P, Q, stream, steps):
    """Secondary object of the N > 1 lines: the distributed shallow-water step (BASELINE configs[3]'s kernel on
    configs[4]'s per-GPU tile), time-loop form -- frame workgroups inside the interior launch, ONE grouped exchange of
    the three new fields (one message per neighbour and direction) hidden behind the interior, joined on the device
    by the next step -- leapfrog rotation of the nine fields, one join closing the loop inside the timed region."""
    steps = max(min(steps, 40), MIN_SECONDARY_LAUNCHES)
    g = make_grid(D, tile * P, tile * Q, alignment)
    pts = {"u": D.GO_U_POINTS, "v": D.GO_V_POINTS, "p": D.GO_T_POINTS}
    names = ["u", "v", "p", "uold", "vold", "pold", "unew", "vnew", "pnew"]
    F = {}
    with torch.cuda.stream(stream):
        for k, name in enumerate(names):
            F[name] = D.r2d_field(g, pts[name[0]])
            D.psy.hash_init(F[name], SEED + k, stream=stream)
            F[name].data.add_(1.0 if name[0] == "p" else -0.5)
        D.psy.halo_exchange_multi([F[n] for n in names[:6]], stream=stream)
    prm = D.psy.shallow_params(1.0e5, 1.0e5, 90.0)
    cur, old, new = [F[n] for n in names[:3]], [F[n] for n in names[3:6]], [F[n] for n in names[6:]]
    # before timing, on every rank: pipelined step + join == plain step + grouped exchange
    chk = [D.r2d_field(g, pts[n[0]]) for n in names[6:]]
    with torch.cuda.stream(stream):
        for x, y in zip(new, chk):
            D.copy_field(x, y, stream=stream)                # cells no step writes (the domain's ring) then agree
        D.psy.invoke_shallow_step(prm, *cur, *old, *chk, stream=stream)
        D.psy.halo_exchange_multi(chk, stream=stream)
        D.psy.invoke_shallow_step_dm_pipelined(prm, *cur, *old, *new, stream=stream)
        D.psy.halo_join(g, stream=stream)
    stream.synchronize()
    w = F["p"].whole
    cut = lambda f: f.data[w.ystart - 1:w.ystop, w.xstart - 1:w.xstop]      # noqa: E731
    ok = torch.tensor([1 if all(torch.equal(cut(x), cut(y)) for x, y in zip(new, chk)) else 0], device="cuda")
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    same = bool(int(ok[0]))
    del chk

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        old, cur, new = cur, new, old
        for _ in range(5):
            D.psy.invoke_shallow_step_dm_pipelined(prm, *cur, *old, *new, stream=stream)
            old, cur, new = cur, new, old
        D.psy.halo_join(g, stream=stream)
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            D.psy.invoke_shallow_step_dm_pipelined(prm, *cur, *old, *new, stream=stream)
            old, cur, new = cur, new, old
        D.psy.halo_join(g, stream=stream)                    # the one join of the loop, inside the timed region
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt[0])
    ms = wall / steps * 1e3
    gbs = 72 * tile * tile / (ms * 1e-3) / 1e9
    out = {"workload": f"shallow-water u/v/h fused step, {tile}x{tile} fp64 per GPU, {P}x{Q} decomposition, "
                       f"grouped RCCL exchange of unew/vnew/pnew (8 directions, one message each) overlapped",
           "tile": tile, "n_gpus": world, "steps": steps, "value": round(tile * tile * world * steps / wall / 1e6, 1),
           "unit": "Mcells/s", "ms_per_step": round(ms, 5), "hbm_gbs_per_gpu": round(gbs, 1),
           "frac_of_hbm_peak_per_gpu": round(gbs / HBM_PEAK_GBS, 4), "scaling": "weak",
           "dm_step_equals_step_plus_exchange": same, "checksum_p": D.field_checksum(cur[2])}
    del F, cur, old, new
    torch.cuda.empty_cache()
    return out


_STAGE = {"name": "start", "t0": time.time()}


def _time_loop(D, torch, dist, world, g, a, b, stream, steps, warmup, step):
    """warm-up, barrier, `steps` x step + ONE join, barrier; max over ranks; returns (seconds, a, b)"""
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            step(b, a, stream=stream)
            a, b = b, a
        D.psy.halo_join(g, stream=stream)
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            step(b, a, stream=stream)
            a, b = b, a
        D.psy.halo_join(g, stream=stream)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt[0])
    return wall, a, b


def peer_transport_dm(D, torch, dist, tile, alignment, world, P, Q, stream, steps, tables=None):
    """Secondary object: the distributed Jacobi step over the PEER TRANSPORT -- the frame
    workgroups of the step launch store into the neighbours' mailboxes over xGMI and raise their arrival flags; no RCCL
    kernel, no side stream -- next to the RCCL form on the SAME grid, same process, same loop.  N > 1: the decomposition of
    the weak-scaling leg; N = 1 with `tables`: one GPU that is its own four neighbours (loop-back), which prices
    everything but the xGMI hop.  Checked first: three steps over the mailboxes == three x (stencil + RCCL edge exchange),
    every bit on every rank; a mismatch, a failed connect or a wait that gave up is reported, not timed."""
    steps = max(steps, MIN_SECONDARY_LAUNCHES)
    L = D._cabi.lib()
    g = make_grid(D, tile * P, tile * Q, alignment)
    a, b, x, y = (D.r2d_field(g, D.GO_T_POINTS) for _ in range(4))
    it = a.internal
    if tables is not None:                                   # loop-back: the plan is made from these tables
        plan = C.c_void_p()
        t = tables(D, it)
        D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
        g._halo_plan = plan
    old_wait = L.dlesm_set_tuning(b"dm_wait_seconds", 60)   # a neighbour that never answers: words after a minute
    L.dlesm_set_tuning(b"dm_peer_exchange", 0)              # the reference exchanges of the self-check go through RCCL

    def agree(flag):
        if world == 1:
            return bool(flag)
        ok = torch.tensor([1 if flag else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return bool(int(ok[0]))

    def leave(obj):
        L.dlesm_set_tuning(b"dm_wait_seconds", old_wait or 600)      # (0 = the key was unset: the default)
        L.dlesm_set_tuning(b"dm_peer", 1)
        L.dlesm_set_tuning(b"dm_peer_exchange", 1)
        L.dlesm_set_tuning(b"mailbox_fences", 1)
        return obj

    err = None
    try:
        D.psy.halo_connect_peers(g)                          # collective
    except Exception as e:                                   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(err is None):
        return leave({"error": "mailboxes not connected on every rank" + (f" (this rank: {err})" if err else "")})
    sp = C.c_void_p(stream.cuda_stream)
    with torch.cuda.stream(stream):
        D.psy.hash_init(a, SEED, stream=stream)
        a.halo_exchange(1, stream=stream)
        D.copy_field(a, b, stream=stream)
        D._cabi.check(L.dlesm_stencil5_autotune_f64(a.device_ptr, b.device_ptr, g.nx, g.ny, it.xstart + 1, it.xstop - 1,
                                                    it.ystart + 1, it.ystop - 1, sp))

    def selfcheck():
        """three steps over the mailboxes (two in the time-loop form, one joined) against three x (stencil + RCCL exchange)"""
        with torch.cuda.stream(stream):
            t1, t2 = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
            for f in (b, y, t1, t2):                         # (a stays the initial state: a second attempt starts from it again)
                D.copy_field(a, f, stream=stream)
            x1, y1, x2, y2 = b, t1, y, t2
            for k in range(3):
                (D.psy.invoke_jacobi5_dm_pipelined if k < 2 else D.psy.invoke_jacobi5_dm)(y1, x1, stream=stream)
                x1, y1 = y1, x1
                D.psy.invoke_jacobi5(y2, x2, stream=stream)
                y2.halo_exchange(1, stream=stream, dirs=D._cabi.DIRS_EDGES_ONLY)
                x2, y2 = y2, x2
        stream.synchronize()
        gave_up = bool(L.dlesm_wait_timed_out(0))
        return gave_up, (not gave_up) and bool(torch.equal(x1.data, x2.data))

    # the library's default: arrival flags as release / acquire pairs (mailbox_fences = 1, round 4)
    fences = 1
    gave_up, same = selfcheck()
    ok = agree(same)
    if not ok:
        if not agree(not gave_up):      # acknowledge, so that the rest of the run still has a library to talk to
            D._cabi.check(L.dlesm_halo_plan_destroy(g._halo_plan))
            g._halo_plan = None
            L.dlesm_wait_timed_out(1)
        return leave({"error": "a wait for a neighbour's arrival flag gave up" if gave_up else
                      "steps over the mailboxes differ from stencil + RCCL exchange on some rank",
                      "equals_stencil_plus_rccl_exchange": False})
    with torch.cuda.stream(stream):
        D.copy_field(a, b, stream=stream)
    res = {}
    for name, peer in (("rccl", 0), ("peer", 1), ("rccl2", 0), ("peer2", 1)):
        L.dlesm_set_tuning(b"dm_peer", peer)
        wall, a, b = _time_loop(D, torch, dist, world, g, a, b, stream, steps, 5, D.psy.invoke_jacobi5_dm_pipelined)
        res[name] = wall / steps * 1e3
    L.dlesm_set_tuning(b"dm_peer", 1)
    plain = float("inf")
    for _ in range(3):                                       # the plain sweep of one tile, no exchange: the yardstick (best of three
        with torch.cuda.stream(stream):                      # passes, as the loops above are the better of two: a 5 ms sample is at
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # the mercy of the clocks)
            for _ in range(3):
                D.psy.invoke_jacobi5(b, a, stream=stream)
            e0.record(stream)
            for _ in range(steps):
                D.psy.invoke_jacobi5(b, a, stream=stream)
                a, b = b, a
            e1.record(stream)
        stream.synchronize()
        plain = min(plain, e0.elapsed_time(e1) / steps)
    # r2d_field%halo_exchange on its own (all eight directions), back to back: RCCL group against the two mailbox launches
    xus = {}
    for name, px in (("rccl", 0), ("peer", 1)):
        L.dlesm_set_tuning(b"dm_peer_exchange", px)
        with torch.cuda.stream(stream):
            for _ in range(5):
                a.halo_exchange(1, stream=stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(50):
                a.halo_exchange(1, stream=stream)
        torch.cuda.synchronize()
        xus[name] = (time.perf_counter() - t0) / 50 * 1e6
    L.dlesm_set_tuning(b"dm_peer_exchange", 0)
    rccl, peer = min(res["rccl"], res["rccl2"]), min(res["peer"], res["peer2"])
    cells = tile * tile * world
    out = {"workload": f"jacobi5 {tile}x{tile} fp64 per GPU, {P}x{Q} decomposition, time-loop form of the distributed step: "
                       "RCCL send/recv group per step vs stores into the neighbours' mailboxes by the frame workgroups",
           "tile": tile, "n_gpus": world, "neighbours": "itself (loop-back: no xGMI hop)" if tables is not None else "xGMI peers",
           "steps": steps, "equals_stencil_plus_rccl_exchange": True,
           "plain_sweep_ms": round(plain, 5),
           "rccl": {"ms_per_step": round(rccl, 5), "value": round(cells / rccl / 1e3, 1), "frac_of_plain_sweep": round(plain / rccl, 4)},
           "peer": {"ms_per_step": round(peer, 5), "value": round(cells / peer / 1e3, 1), "frac_of_plain_sweep": round(plain / peer, 4)},
           "unit": "Mcells/s", "speedup": round(rccl / peer, 3), "mailbox_fences": fences,
           "halo_exchange_us": {"rccl": round(xus["rccl"], 1), "peer": round(xus["peer"], 1),
                                "what": "r2d_field%halo_exchange of one field, eight directions, 50 back to back, this rank's wall clock"}}
    if L.dlesm_wait_timed_out(0):
        out["error"] = "a wait gave up during the timed loops"
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()                                       # nobody frees a mailbox a neighbour may still be mapping
    D._cabi.check(L.dlesm_halo_plan_destroy(g._halo_plan))
    g._halo_plan = None
    del a, b, x, y
    torch.cuda.empty_cache()
    return leave(out)


def loopback_tables(D, it):
    """message tables of a depth-1 exchange in which rank 0 is its own eight neighbours (a periodic wrap onto itself)"""
    t = D._cabi.CommTables()
    xl, xh, yl, yh = it.xstart, it.xstop, it.ystart, it.ystop
    msgs = [(2, xh, yl, it.xstart - 1, yl, 1, it.ny), (1, xl, yl, it.xstop + 1, yl, 1, it.ny),
            (4, xl, yh, xl, it.ystart - 1, it.nx, 1), (3, xl, yl, xl, it.ystop + 1, it.nx, 1),
            (6, xh, yh, it.xstart - 1, it.ystart - 1, 1, 1), (5, xl, yl, it.xstop + 1, it.ystop + 1, 1, 1),
            (7, xl, yh, it.xstop + 1, it.ystart - 1, 1, 1), (8, xh, yl, it.xstart - 1, it.ystop + 1, 1, 1)]
    t.nsend = t.nrecv = len(msgs)
    for k, (d, isrc, jsrc, ides, jdes, nx, ny) in enumerate(msgs):
        t.dirsend[k] = t.dirrecv[k] = d
        t.destination[k] = t.source[k] = 0
        t.isrcsend[k], t.jsrcsend[k], t.idessend[k], t.jdessend[k] = isrc, jsrc, ides, jdes
        t.nxsend[k], t.nysend[k] = nx, ny
        t.isrcrecv[k], t.jsrcrecv[k], t.idesrecv[k], t.jdesrecv[k] = isrc, jsrc, ides, jdes
        t.nxrecv[k], t.nyrecv[k] = nx, ny
    return t


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; this program's stdout carries ONE JSON line.
    File-descriptor level, because the banner comes from C."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def stage(rank, world, name):
    """N > 1 only: one stderr line per stage and rank, so that a failed or hung multi-GPU run (which
    cannot be rehearsed on the one-GPU development box) says where it stopped"""
    _STAGE["name"] = name
    if world > 1:
        print(f"[bench rank {rank}/{world} +{time.time() - _STAGE['t0']:6.1f}s] {name}", file=sys.stderr, flush=True)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # a run that hangs before the headline exists must not sit there silently until the launcher's
        # own limit: after 900 s every rank reports the stage it is stuck in and leaves non-zero
        import threading

        def stuck():
            print(f"[bench rank {rank}/{world}] NO PROGRESS: still in stage '{_STAGE['name']}' after 900 s; giving up",
                  file=sys.stderr, flush=True)
            os._exit(6)

        killer = threading.Timer(900.0, stuck)
        killer.daemon = True
        killer.start()
    stage(rank, world, "process group init")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # DLESM_TRANSPORT=mailbox: the library's mode without a communication library -- the ranks may
    # then share a GPU (RCCL refuses that), which is how the N > 1 path of this program is rehearsed on a one-GPU box;
    # torch's own group (barriers, the max over ranks) is gloo in that case
    mailbox = os.environ.get("DLESM_TRANSPORT") == "mailbox"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1 or args.force_dm_leg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with stdout_to_stderr():
            if mailbox:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local))
            dist.barrier()                       # torch creates its communicator at the first collective

    import dl_esm_inf_amd as D
    L = D._cabi.lib()
    if args.rows is not None:
        L.dlesm_set_tuning(b"j5_rows", args.rows)
    if args.variant is not None:
        L.dlesm_set_tuning(b"j5_variant", args.variant)
    for kv in args.tune:
        k, v = kv.split("=")
        L.dlesm_set_tuning(k.encode(), int(v))     # returns the previous value
    if args.dm_form == "safe":
        L.dlesm_set_tuning(b"dm_safe", 1)
    D._cabi.check(L.dlesm_init(local))             # after the knobs: the side stream's priority is one of them
    os.environ["DL_ESM_ALIGNMENT"] = str(args.alignment)
    stage(rank, world, "RCCL communicator (dlesm_comm_init)")
    with stdout_to_stderr():
        D.parallel_init(rank, world, transport="mailbox" if mailbox else None)
    stage(rank, world, "grid + fields + first halo exchange")

    # global domain: what go_decompose will cut into `world` tiles of tile x tile
    small = int(math.isqrt(world))
    while world % small:
        small -= 1
    P, Q = small, world // small
    grid = D.grid_type(D.GO_ARAKAWA_C, (D.GO_BC_EXTERNAL, D.GO_BC_EXTERNAL, D.GO_BC_NONE), D.GO_OFFSET_NE)
    grid.decompose(args.tile * P, args.tile * Q)
    assert (grid.decomp.nx, grid.decomp.ny) == (P, Q)
    D.grid_init(grid, 1.0, 1.0)
    a, b = D.r2d_field(grid, D.GO_T_POINTS), D.r2d_field(grid, D.GO_T_POINTS)
    it = a.internal
    assert it.nx == args.tile and it.ny == args.tile
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        D.psy.hash_init(a, SEED)             # whole region incl. the fixed boundary ring
        D.copy_field(a, b)                   # same ring in both ping-pong buffers
        a.halo_exchange(1, stream=stream)    # `in` starts with valid halos
    stream.synchronize()

    # N > 1, --dm-form timeloop: the exchange of step k is joined by step k+1's frame workgroups on the device (flag poll,
    # one agent-scope acquire, barrier), the caller's stream carries one launch per step; ONE join closes the loop.
    # joined / safe: every step joins its own exchange (the closing join then finds nothing pending).
    dm_step = D.psy.invoke_jacobi5_dm_pipelined if args.dm_form == "timeloop" else D.psy.invoke_jacobi5_dm
    step = dm_step if world > 1 else D.psy.invoke_jacobi5
    stage(rank, world, "planning call")
    planned = False
    if not args.no_plan:
        # planning, outside the timed region (like an FFT plan): the library times its launch shapes for
        # this geometry on the bench's own arrays and keeps the fastest; results do not depend on the shape.
        # N > 1: the interior box of the distributed step is what the bulk of the time goes to.
        with torch.cuda.stream(stream):
            if world == 1:
                D.psy.autotune_jacobi5(b, a, stream=stream)
            else:
                D._cabi.check(L.dlesm_stencil5_autotune_f64(a.device_ptr, b.device_ptr, grid.nx, grid.ny,
                                                            it.xstart + 1, it.xstop - 1, it.ystart + 1, it.ystop - 1,
                                                            C.c_void_p(stream.cuda_stream)))
            D.copy_field(a, b, stream=stream)            # b back to its starting state
        stream.synchronize()
        planned = True
    # the kernel the planning call settled on (rows per tile is part of the plan), for the roofline object
    shape = D.psy.planned_shape_jacobi5(b)
    nt = 2 if shape[3] else 0
    j5_kernel = f"jacobi5_tile<2,{shape[2] or 2},{nt}>" if world == 1 else f"jacobi5_tile_framed<2,2,{nt}>"
    fused = args.fused
    if fused != 1:
        if world > 1 or not 2 <= fused <= 8 or args.steps % fused:
            raise SystemExit("bench.py --fused T: 1 GPU, T in 2..8, --steps a multiple of T")

        def step(o, i, stream=None):                         # noqa: F811  (one launch = `fused` time steps)
            D.psy.invoke_jacobi5_multi(o, i, fused, stream=stream)
    launches, warm_launches = args.steps // fused, -(-args.warmup // fused)

    # N > 1: before timing anything, `--selfcheck-steps` (50) distributed steps in the TIMED form, issued back to back exactly
    # as the timed loop issues them -- so that the halo lines each step reads are as warm in L1 / L2 as they will be there;
    # a three-step check from a cold start cannot see a hand-over that only fails warm -- must reproduce, bit for bit on every
    # rank, as many plain "stencil, then halo exchange" steps (kernel boundaries only) from the same state
    selfcheck = None
    dm_safe_fallback = False
    a0 = None
    if world > 1 or args.force_dm_leg:
        a0 = D.r2d_field(grid, D.GO_T_POINTS)                # the initial state, halos included: start of the dm_safe re-run below
        with torch.cuda.stream(stream):
            D.copy_field(a, a0, stream=stream)
        nchk = max(3, args.selfcheck_steps)

        def run_selfcheck():
            x1, y1 = D.r2d_field(grid, D.GO_T_POINTS), D.r2d_field(grid, D.GO_T_POINTS)
            x2, y2 = D.r2d_field(grid, D.GO_T_POINTS), D.r2d_field(grid, D.GO_T_POINTS)
            with torch.cuda.stream(stream):
                for f in (x1, y1, x2, y2):
                    D.copy_field(a, f, stream=stream)
                for k in range(nchk):
                    dm_step(y1, x1, stream=stream)
                    x1, y1 = y1, x1
                D.psy.halo_join(grid, stream=stream)
                for k in range(nchk):
                    D.psy.invoke_jacobi5(y2, x2, stream=stream)
                    y2.halo_exchange(1, stream=stream, dirs=D._cabi.DIRS_EDGES_ONLY)   # what the 5-point step exchanges
                    x2, y2 = y2, x2
            stream.synchronize()
            ok = torch.tensor([1 if torch.equal(x1.data, x2.data) else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            del x1, y1, x2, y2
            torch.cuda.empty_cache()
            return bool(int(ok[0]))

        stage(rank, world, f"self-check: {nchk} distributed steps ({args.dm_form}) == {nchk} x (stencil + exchange)")
        selfcheck = run_selfcheck()
        if not selfcheck and args.dm_form != "safe":
            # The one-launch / time-loop forms hand over through device flags.  If they do not reproduce
            # stencil + exchange on this machine, fall back -- on every rank, the decision is collective -- to the
            # conservative forms (events and kernel boundaries only, DLESM_DM_SAFE) and check again; the line then says so
            # and the process leaves non-zero AFTER printing it.
            stage(rank, world, "self-check FAILED in the one-launch forms: falling back to DLESM_DM_SAFE")
            L.dlesm_set_tuning(b"dm_safe", 1)
            dm_safe_fallback = True
            selfcheck = run_selfcheck()
        if not selfcheck:
            raise SystemExit("bench.py: distributed step differs from stencil + exchange, in the conservative form too")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stage(rank, world, "warm-up + timed steps")
    with torch.cuda.stream(stream):
        for _ in range(warm_launches):
            step(b, a, stream=stream)
            a, b = b, a
        if world > 1:
            D.psy.halo_join(grid, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(launches):
            step(b, a, stream=stream)
            a, b = b, a
        if world > 1:
            D.psy.halo_join(grid, stream=stream)             # inside the timed region
        e1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)

    if world > 1:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])
    stage(rank, world, "global checksum (ncclAllReduce)")
    checksum = D.field_checksum(a)
    # N > 1: the SAME warm-up + timed steps once more from the same initial state in the conservative form (DLESM_DM_SAFE:
    # own frame launch, events, unpack into the field -- kernel boundaries only), untimed: the field the timed loop left
    # behind must equal it bit for bit on every rank, and so must the global checksum
    dm_rerun = None
    if a0 is not None:
        stage(rank, world, "re-run of the timed steps in DLESM_DM_SAFE, compared with what the timed loop left")
        was_safe = L.dlesm_set_tuning(b"dm_safe", 1)
        x, y = a0, D.r2d_field(grid, D.GO_T_POINTS)
        with torch.cuda.stream(stream):
            D.copy_field(x, y, stream=stream)
            for _ in range(warm_launches + launches):
                D.psy.invoke_jacobi5_dm(y, x, stream=stream)
                x, y = y, x
        stream.synchronize()
        it_ = a.internal
        cut = lambda f: f.data[it_.ystart - 2:it_.ystop + 1, it_.xstart - 2:it_.xstop + 1]      # noqa: E731  (internal + halo ring)
        ok = torch.tensor([1 if torch.equal(cut(x), cut(a)) else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        cs_safe = D.field_checksum(x)
        dm_rerun = {"steps": warm_launches + launches, "fields_equal_on_every_rank": bool(int(ok[0])),
                    "checksum_dm_safe": cs_safe, "checksums_equal": cs_safe == checksum}
        L.dlesm_set_tuning(b"dm_safe", was_safe)
        del x, y, a0
        a0 = None
        torch.cuda.empty_cache()
    stage(rank, world, "secondary legs")

    cells_step = args.tile * args.tile * world
    value = cells_step * args.steps / wall / 1e6
    launch_ms = ev_ms / launches
    achieved =BYTES_PER_CELL * args.tile * args.tile / (launch_ms * 1e-3) / 1e9   # per GPU, GB/s
    out = {
        "metric": "stencil Mcells/s", "value": round(value, 1), "unit": "Mcells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"jacobi5 {args.tile}x{args.tile} fp64 T-field per GPU "
                               f"({baseline_config(args.tile, world)}; NE offset, external BCs, fixed boundary ring)",
                   "tile": args.tile, "decomposition": f"{P}x{Q}",
                   "global": [args.tile * P, args.tile * Q], "DL_ESM_ALIGNMENT": args.alignment,
                   "ld": grid.nx,
                   "halo_exchange": (("MAILBOX MODE (DLESM_TRANSPORT=mailbox: no RCCL; the frame workgroups store into the "
                                      "neighbours' mailboxes) -- a rehearsal when the ranks share one GPU; " if mailbox else
                                      "rccl send/recv of the four edges, overlapped; ") +
                                     ("CONSERVATIVE form (DLESM_DM_SAFE: own frame launch, event joins) after a failed "
                                      "self-check of the one-launch forms" if dm_safe_fallback else
                                      {"timeloop": "time-loop form: step k+1's frame workgroups join step k's exchange on the "
                                                   "device -- flag poll, one agent-scope acquire + s_waitcnt, barrier, then the "
                                                   "halo loads" + ("" if "dm_acquire=0" not in args.tune else
                                                                   " -- ACQUIRE SWITCHED OFF (--tune dm_acquire=0): not a guide-valid hand-over"),
                                       "joined": "joined form: every step joins its exchange on the caller's stream (flag-wait kernel)",
                                       "safe": "DLESM_DM_SAFE: own frame launch, events, unpack into the field"}[args.dm_form]))
                   if world > 1 else "none (1 tile)",
                   "launch_shape": "planned (dlesm_stencil5_autotune_f64, before the warm-up)" if planned else "rule",
                   "planned_waves_tiles_rows_ntstores": list(shape)},
        "hbm_gbs_per_gpu": round(achieved, 1),
        "checksum": checksum, "dm_step_equals_stencil_plus_exchange": selfcheck,
        "dm_form": args.dm_form if (world > 1 or args.force_dm_leg) else None,
        "dm_selfcheck_steps": max(3, args.selfcheck_steps) if selfcheck is not None else None,
        "dm_safe_fallback": dm_safe_fallback, "dm_safe_rerun": dm_rerun,
        # (the contract's vocabulary for the headline object is "hbm" | "mfma"; a tile small enough to ping-pong inside
        #  the 256 MiB Infinity Cache -- never the BASELINE headline -- is flagged beside it)
        "roofline": {"bound": "hbm", "infinity_cache_resident": grid.nx * grid.ny * 8 < (150 << 20),
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic_for(args.tile, args.alignment, fused), "traffic_source": TRAFFIC_SOURCE,
                     "kernel": j5_kernel if fused == 1 else f"jacobi5xt_tile<{fused},{XT_ROWS[fused]},dpp>",
                     "launch_ms": round(launch_ms, 5),
                     "algorithmic_bytes_per_launch": BYTES_PER_CELL * args.tile * args.tile},
    }
    secondary = fused == 1 and not args.force_dm_leg
    failed = []                                              # names of secondary legs that raised

    def guarded(name, fn):
        """a secondary leg must never cost the headline line: its error text is reported instead"""
        try:
            out[name] = fn()
        except Exception as e:                               # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            failed.append(name)

    if fused > 1:
        out["config"]["fused_steps_per_launch"] = fused
        out["roofline"]["note"] = (f"one launch advances {fused} time steps; bytes are per launch, so Mcells/s "
                                   f"exceeds what 16 B/cell/step allows at this bandwidth")
    elif world == 1 and not args.force_dm_leg:
        if not args.no_temporal_blocking:
            guarded("temporal_blocking", lambda: temporal_blocking(D, torch, grid, a, stream, args.steps, args.tile))
    else:
        # N > 1 (or its 1-rank rehearsal): every secondary leg contains collectives, so a rank that
        # fails would leave the others waiting.  If a leg raises, or the legs have not finished in
        # 180 s, rank 0 prints the line with what it has and EVERY rank leaves with a non-zero status:
        # a hung or failed leg must not look like a clean run.
        import threading

        def bail(why, code):
            if rank == 0:
                out.setdefault("cpu_baseline", None)
                out["secondary_legs_error"] = why
                print(json.dumps(out), flush=True)
            os._exit(code)

        dog = threading.Timer(180.0, bail, args=("secondary legs did not finish in 180 s", 3))
        dog.daemon = True
        dog.start()
        legs_t0 = time.perf_counter()
        try:
            if not args.no_weak_tile and args.tile != WEAK_TILE:
                stage(rank, world, "secondary leg: 8192^2 weak-scaling tile")
                out["weak_scaling_tile"] = weak_scaling_tile(D, torch, dist, WEAK_TILE, args.alignment, world, P, Q,
                                                             stream, args.steps, dm_form=args.dm_form)
            if not args.no_temporal_blocking:
                stage(rank, world, "secondary leg: fused 8-step distributed form")
                out["temporal_blocking"] = temporal_blocking_dm(D, torch, dist, args.tile, P, Q, stream, args.steps)
            if not args.no_shallow:
                stage(rank, world, "secondary leg: distributed shallow-water step, 8192^2 per GPU")
                out["shallow_water_dm"] = shallow_water_dm(D, torch, dist, min(WEAK_TILE, args.tile), args.alignment, world,
                                                           P, Q, stream, args.steps)
            spent = torch.tensor([time.perf_counter() - legs_t0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(spent, op=dist.ReduceOp.MAX)     # (every rank takes the same decision)
            if not args.no_peer and not os.environ.get("DLESM_BENCH_NO_PEER") and not mailbox and float(spent[0]) > 100.0:
                out["peer_transport"] = {"skipped": f"the other secondary legs took {float(spent[0]):.0f} s of the 180 s they share"}
            elif not args.no_peer and not os.environ.get("DLESM_BENCH_NO_PEER") and not mailbox:
                # LAST: the one leg whose transport no earlier run has exercised between GPUs
                stage(rank, world, "secondary leg: peer transport (mailboxes) next to RCCL, 8192^2 per GPU")
                try:                                         # an extra: what goes wrong in it is reported, not fatal
                    out["peer_transport"] = peer_transport_dm(D, torch, dist, min(WEAK_TILE, args.tile), args.alignment,
                                                              world, P, Q, stream, args.steps)
                except Exception as e:                       # noqa: BLE001
                    out["peer_transport"] = {"error": f"{type(e).__name__}: {e}"}
        except Exception as e:                               # noqa: BLE001
            dog.cancel()
            bail(f"{type(e).__name__}: {e}", 4)              # the other ranks may be stuck in a collective
        dog.cancel()
    if world == 1 and secondary:
        def headline_ceiling():
            # one array read + one written, the headline's field shape, same process (two fresh arrays)
            x, y = (torch.empty((grid.ny, grid.nx), dtype=torch.float64, device="cuda") for _ in range(2))
            x.copy_(a.data)
            torch.cuda.synchronize()                         # (the copy ran on torch's current stream, the sweeps run on `stream`)
            cc = copy_ceiling(D, torch, stream, [x], [y], grid.nx * grid.ny)
            out["roofline"]["frac_of_copy_ceiling"] = round(achieved / cc["best_gbs"], 4)
            return cc
        guarded("copy_ceiling", headline_ceiling)
        torch.cuda.empty_cache()
        if not args.no_shallow:
            guarded("shallow_water", lambda: shallow_water(
                D, torch, stream, args.alignment, tile=min(8192, args.tile), steps=min(40, args.steps),
                cpu_seconds=0.0 if args.no_cpu_baseline else min(4.0, args.cpu_seconds), plan=not args.no_plan))
        if not args.no_weak_tile and args.tile != WEAK_TILE:
            guarded("weak_scaling_tile", lambda: weak_scaling_tile(D, torch, None, WEAK_TILE, args.alignment, 1, 1, 1,
                                                                   stream, args.steps))
        if not args.no_peer and not os.environ.get("DLESM_BENCH_NO_PEER"):
            def dm_loopback():
                # one GPU that is its own four neighbours: the small tile the exchange is hardest to hide behind
                with stdout_to_stderr():
                    D.parallel_init(0, 1, use_rccl=True)
                try:
                    return peer_transport_dm(D, torch, None, min(4096, args.tile), args.alignment, 1, 1, 1, stream,
                                             max(args.steps, 100), tables=loopback_tables)
                finally:
                    D.parallel_finalise()
            # an extra, not one of BASELINE's configurations: if it cannot run here (e.g. no RCCL for the loop-back
            # communicator) the line says why and the run still ends with status 0
            try:
                out["dm_loopback"] = dm_loopback()
            except Exception as e:                           # noqa: BLE001
                out["dm_loopback"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_configs:
            # the other BASELINE Jacobi configurations, timed in the same process
            legs = [(4096, 64), (16384, 1), (4096, 1)]
            out["configs"] = []
            for (t, al) in legs:
                if (t, al) == (args.tile, args.alignment):
                    continue
                try:
                    out["configs"].append(jacobi_config(D, torch, stream, t, al, args.steps))
                except Exception as e:                       # noqa: BLE001
                    out["configs"].append({"tile": t, "DL_ESM_ALIGNMENT": al, "error": f"{type(e).__name__}: {e}"})
                    failed.append(f"configs[{t},A{al}]")
    if rank == 0 and not args.no_cpu_baseline:
        # N > 1: a shorter sample (the other ranks wait at the barrier below), same tile, rank 0's cores
        host = a.get_data()
        out["cpu_baseline"] = cpu_baseline(host, grid.nx, it.box(), args.cpu_seconds if world == 1
                                           else min(args.cpu_seconds, 5.0))
    elif rank == 0:
        out["cpu_baseline"] = None
    if failed:
        out["secondary_legs_error"] = "failed: " + ", ".join(failed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    stage(rank, world, "shutdown")
    if world > 1:
        killer.cancel()
    if world > 1 or args.force_dm_leg:
        dist.barrier()
        D.parallel_finalise()
        dist.destroy_process_group()
    if failed:
        sys.exit(5)
    if dm_safe_fallback:
        sys.exit(7)
    if dm_rerun and not (dm_rerun["fields_equal_on_every_rank"] and dm_rerun["checksums_equal"]):
        sys.exit(8)                                          # the timed loop and its DLESM_DM_SAFE re-run disagree: the line says so


if __name__ == "__main__":
    main()
