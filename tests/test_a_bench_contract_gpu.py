"""GPU test (-m gpu): bench.py's output contract on a small tile, and a 1-rank rehearsal of the
leg that only runs when N > 1 (the fused distributed step on a halo_width-4 decomposition)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _bench(*flags):
    # bench.py runs as a child process; this file sorts first so that the pytest process has not
    # touched the GPU yet when it forks (the GPU boxes refuse an exec from a process that has)
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tile", "2048", "--steps", "8",
                        "--warmup", "2", "--cpu-seconds", "0.5", *flags], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]       # ONE line on stdout, and it is the JSON
    return json.loads(lines[0])


def test_bench_line_and_secondary_legs():
    d = _bench()
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["unit"] == "Mcells/s" and d["dtype"] == "f64"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["cpu_baseline"]["fortran_psy_loops_value"] > 0, d["cpu_baseline"]
    sw = d["shallow_water"]
    assert "error" not in sw and sw["value"] > 0 and sw["roofline"]["algorithmic_bytes_per_cell"] == 72, sw
    assert sw["cpu_baseline"]["gpu_first_step_equals_oracle_on_slab"] is True and sw["cpu_baseline"]["cores"] >= 1, sw
    assert sw["cpu_baseline"]["fortran_loops_equal_c_oracle"] is True and sw["cpu_baseline"]["single_core_value"] > 0, sw
    # round 3: the un-fused GOcean kernel sequence beside the fused step, the same-run copy ceilings, one-launch periodic step
    un = sw["unfused"]
    assert "error" not in un and un["bit_identical_to_fused_step"] is True and un["roofline"]["algorithmic_bytes_per_cell"] == 224, un
    assert set(un["per_kernel"]) == {"cu", "cv", "z", "h", "unew", "vnew", "pnew"} and un["time_smooth"]["ms"] > 0, un
    ts = sw["with_time_smooth"]
    assert "error" not in ts and ts["bit_identical_to_step_plus_time_smooth"] is True and ts["roofline"]["algorithmic_bytes_per_cell"] == 96, ts
    assert sw["copy_ceiling"]["best_gbs"] > 0 and 0 < sw["roofline"]["frac_of_copy_ceiling"] < 2, sw
    tx2 = ts["two_steps_per_launch"]         # ... and two whole FILTERED time steps per launch
    assert "error" not in tx2 and tx2["bit_identical_to_two_one_launch_filtered_steps"] is True and tx2["value"] > 0, tx2
    x2 = sw["two_steps_per_launch"]          # round 4: two leapfrog steps per launch, 48 B/cell/step
    assert "error" not in x2 and x2["bit_identical_to_two_single_steps"] is True and x2["value"] > 0, x2
    assert x2["roofline"]["algorithmic_bytes_per_cell_per_step"] == 48 and x2["copy_ceiling"]["best_gbs"] > 0, x2
    assert sw["sw_offset_periodic"]["one_launch_equals_step_plus_halo_copies"] is True, sw["sw_offset_periodic"]
    pp = sw["sw_offset_periodic"]["two_steps_per_launch"]
    assert "error" not in pp and pp["bit_identical_to_two_one_launch_steps_fields_and_halos"] is True and pp["value"] > 0, pp
    pts = sw["sw_offset_periodic"]["with_time_smooth"]       # round 4: the benchmark's filtered loop, one and two steps per launch
    assert "error" not in pts and pts["two_steps_per_launch"]["bit_identical_to_two_one_launch_steps_fields_and_halos"] is True, pts
    assert pts["two_steps_per_launch"]["value"] > 0 and pts["one_launch_per_step"]["value"] > 0, pts
    assert d["copy_ceiling"]["best_gbs"] > 0 and 0 < d["roofline"]["frac_of_copy_ceiling"] < 2, d["copy_ceiling"]
    lb = d["dm_loopback"]        # round 3: the distributed step over the peer transport beside the RCCL form, loop-back
    assert "error" not in lb and lb["equals_stencil_plus_rccl_exchange"] is True, lb
    assert lb["peer"]["value"] > 0 and lb["rccl"]["value"] > 0 and 0.3 < lb["peer"]["frac_of_plain_sweep"] < 1.2, lb
    tb = d["temporal_blocking"]
    assert tb["fused_steps"] == 8 and tb["bit_identical_to_single_steps"] is True and tb["value"] > 0
    assert tb["steps"] >= 24 * 8                       # secondary legs time >= 24 launches whatever --steps is
    w = d["weak_scaling_tile"]
    assert w["tile"] == 8192 and w["n_gpus"] == 1 and w["value"] > 0 and w["steps"] >= 24, w
    cfgs = {(c["tile"], c["DL_ESM_ALIGNMENT"]): c for c in d["configs"]}
    assert set(cfgs) == {(4096, 64), (16384, 1), (4096, 1)} and all("error" not in c for c in cfgs.values()), cfgs
    assert "configs[1]" in cfgs[(4096, 64)]["workload"] and "configs[2]" in cfgs[(16384, 1)]["workload"]
    assert "traffic_source" in d["roofline"] and "secondary_legs_error" not in d
    f = _bench("--fused", "4", "--no-cpu-baseline")
    assert f["config"]["fused_steps_per_launch"] == 4 and f["value"] > 0


def test_reference_default_alignment():
    """DL_ESM_ALIGNMENT unset is the reference's default (grid_mod.f90:349-369): ld = N + 3, odd for even N, so nx * ny is
    odd -- round 3's `bench.py --alignment 1` left with status 5 because the copy-ceiling legs handed that odd count to a
    sweep of 16-byte elements.  Every leg that times a copy ceiling must come back, headline and shallow-water alike."""
    d = _bench("--alignment", "1", "--no-cpu-baseline", "--no-temporal-blocking", "--no-configs", "--no-weak-tile", "--no-peer")
    assert d["config"]["ld"] % 2 == 1 and "secondary_legs_error" not in d, d.get("secondary_legs_error")
    assert d["copy_ceiling"]["best_gbs"] > 0 and 0 < d["roofline"]["frac_of_copy_ceiling"] < 2, d["copy_ceiling"]
    sw = d["shallow_water"]
    assert "error" not in sw and sw["copy_ceiling"]["best_gbs"] > 0, sw
    assert all("error" not in k and k["frac_of_copy_ceiling"] > 0 for k in sw["unfused"]["per_kernel"].values()), sw["unfused"]
    assert "error" not in sw["with_time_smooth"] and sw["with_time_smooth"]["copy_ceiling"]["best_gbs"] > 0, sw["with_time_smooth"]


def test_rehearsal_of_the_multi_gpu_secondary_leg():
    d = _bench("--force-dm-leg", "--no-cpu-baseline")
    assert d["dm_form"] == "timeloop" and d["dm_selfcheck_steps"] == 50 and d["dm_step_equals_stencil_plus_exchange"] is True, d
    assert d["dm_safe_rerun"]["fields_equal_on_every_rank"] is True and d["dm_safe_rerun"]["checksums_equal"] is True, d["dm_safe_rerun"]
    tb = d["temporal_blocking"]
    assert "error" not in tb, tb
    assert tb["halo_depth"] == 8 and tb["bit_identical_to_single_steps_plus_exchange"] is True
    w = d["weak_scaling_tile"]                          # the 8192^2 object every N > 1 line carries
    assert w["tile"] == 8192 and w["value"] > 0 and "secondary_legs_error" not in d, d
    pt = d["peer_transport"]                            # ... and the mailbox transport next to RCCL (no messages at 1 rank)
    assert "error" not in pt and pt["equals_stencil_plus_rccl_exchange"] is True and pt["peer"]["value"] > 0, pt
    sw = d["shallow_water_dm"]                          # the distributed shallow-water leg of the N > 1 lines
    assert sw["value"] > 0 and sw["dm_step_equals_step_plus_exchange"] is True, sw


@pytest.mark.parametrize("world,form", [(2, "timeloop"), (4, "timeloop"), (2, "joined"), (2, "safe")])
def test_multi_rank_bench_in_mailbox_mode(world, form):
    """`bench.py --gpus N` launched as the driver launches it (one process per rank, RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* in the environment), N = 2 and 4 on the ONE GPU of this box: DLESM_TRANSPORT=mailbox takes the library to its
    mode without a communication library (RCCL refuses ranks that share a device), torch's group is gloo.  Every N > 1
    code path of the program runs -- decomposition, the self-check of three distributed steps against stencil + exchange
    on every rank, the timed time loop with its one join, max over ranks, and the secondary legs (8192^2 weak-scaling
    tile, fused 8-step form on depth-8 halos, distributed shallow-water step) with their own checks.  The rates mean
    nothing (the ranks share a GPU); the checks and the line's shape do."""
    import socket
    import torch
    assert not torch.cuda.is_initialized(), "run this file before any in-process GPU test"
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DLESM_TRANSPORT="mailbox", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--tile", "2048",
                                       "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--dm-form", form], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{err[-3000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), outs[0][0][-2000:]
    assert all(not o[0].strip() for o in outs[1:]), "only rank 0 prints"
    d = json.loads(lines[0])
    assert KEYS - {"cpu_baseline"} <= set(d) and d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0, d
    assert d["dm_step_equals_stencil_plus_exchange"] is True and d["dm_safe_fallback"] is False, d
    # round 4: the self-check is 50 back-to-back steps of the timed form; the timed loop's field == a DLESM_DM_SAFE re-run
    assert d["dm_form"] == form and d["dm_selfcheck_steps"] == 50, d
    assert d["dm_safe_rerun"]["fields_equal_on_every_rank"] is True and d["dm_safe_rerun"]["checksums_equal"] is True, d["dm_safe_rerun"]
    assert "MAILBOX MODE" in d["config"]["halo_exchange"] and d["config"]["decomposition"] in ("1x2", "2x2"), d["config"]
    assert "secondary_legs_error" not in d, d.get("secondary_legs_error")
    w, tb, sw = d["weak_scaling_tile"], d["temporal_blocking"], d["shallow_water_dm"]
    assert w["n_gpus"] == world and w["dm_step_equals_stencil_plus_exchange"] is True and w["value"] > 0, w
    assert tb["bit_identical_to_single_steps_plus_exchange"] is True, tb
    assert sw["dm_step_equals_step_plus_exchange"] is True and sw["value"] > 0, sw
