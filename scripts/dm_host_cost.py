#!/usr/bin/env python3
"""Host issue time per call against device time per step: plain sweep, distributed step (time-loop and joined forms), exchange.
   python scripts/dm_host_cost.py [tile]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, dl_esm_inf_amd as D
from dm_overhead import loopback_tables
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"] = "64"
D.parallel_init(0, 1, use_rccl=True)
g = D.grid_type(D.GO_ARAKAWA_C, (1, 1, 2), D.GO_OFFSET_NE); g.decompose(tile, tile); D.grid_init(g, 1.0, 1.0)
a, b = D.r2d_field(g, D.GO_T_POINTS), D.r2d_field(g, D.GO_T_POINTS)
it = a.internal; t = loopback_tables(D, it); plan = C.c_void_p()
D._cabi.check(L.dlesm_halo_plan_create(C.byref(t), g.nx, g.ny, C.byref(plan)))
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream); box = it.box()
fns = {"plain": lambda x, y: L.dlesm_stencil5_f64(x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp),
       "pipelined": lambda x, y: L.dlesm_jacobi5_step_dm_pipelined(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp),
       "joined": lambda x, y: L.dlesm_jacobi5_step_dm(plan, x.device_ptr, y.device_ptr, g.nx, g.ny, *box, sp),
       "exchange": lambda x, y: L.dlesm_halo_exchange_f64(plan, y.device_ptr, 0x1F, sp)}
for name, fn in fns.items():
    x, y = a, b
    for _ in range(20): fn(x, y); x, y = y, x
    L.dlesm_halo_plan_join(plan, sp); s.synchronize()
    # host issue cost: a long kernel first so that the GPU never starves the queue -> we time the host only
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): fn(x, y); x, y = y, x
    t1 = time.perf_counter()
    L.dlesm_halo_plan_join(plan, sp); s.synchronize()
    t2 = time.perf_counter()
    print(f"tile {tile} {name}: host issue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step", flush=True)
