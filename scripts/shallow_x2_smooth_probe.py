import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dl_esm_inf_amd as D
L = D._cabi.lib()
n = 8192
torch.cuda.set_device(0)
os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE)
g.decompose(n, n)
D.grid_init(g, 1.0, 1.0)
F = [D.r2d_field(g, D.GO_T_POINTS) for _ in range(12)]
for k, f in enumerate(F):
    D.psy.hash_init(f, 77 + k % 6)
    f.data.mul_(0.1)
    f.data.add_(1.0 if k % 3 == 2 else -0.05)
prm = D.psy.shallow_params(1.0e5, 1.0e5, 20.0)
s = torch.cuda.Stream()
def timed(fn, reps=30):
    with torch.cuda.stream(s):
        for _ in range(4): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): fn()
        e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / reps
res = {}
res["one_launch_filtered_step_ms"] = timed(lambda: D.psy.invoke_shallow_step_smooth(prm, 0.001, *F[:9], stream=s))
for rows in (4, 2):
    for nt in (2, 6):
        L.dlesm_set_tuning(b"sw_x2_rows", rows); L.dlesm_set_tuning(b"sw_x2_nt", nt)
        res[f"smooth_x2 R{rows} nt{nt} ms_per_launch"] = timed(lambda: D.psy.invoke_shallow_step_smooth_x2(prm, 0.001, *F, stream=s))
        res[f"plain_x2 R{rows} nt{nt} ms_per_launch"] = timed(lambda: D.psy.invoke_shallow_step_x2(prm, *F, stream=s))
print(json.dumps(res, indent=1))
