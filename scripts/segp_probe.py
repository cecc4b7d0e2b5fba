import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch, dl_esm_inf_amd as D
L = D._cabi.lib(); torch.cuda.set_device(0); os.environ["DL_ESM_ALIGNMENT"]="64"; D.parallel_init(0,1)
tile=16384
g = D.grid_type(D.GO_ARAKAWA_C,(1,1,2),D.GO_OFFSET_NE); g.decompose(tile,tile); D.grid_init(g,1.0,1.0)
a,b = D.r2d_field(g,D.GO_T_POINTS), D.r2d_field(g,D.GO_T_POINTS); it=a.internal
s=torch.cuda.Stream(); sp=C.c_void_p(s.cuda_stream); cells=tile*tile
D.psy.hash_init(a,1,stream=s)
send=torch.zeros(cells,dtype=torch.float64,device="cuda"); glob=torch.zeros(cells,dtype=torch.float64,device="cuda")
res=torch.zeros(1,dtype=torch.float64,device="cuda"); pd=g.decomp
def timed(name, fn, bpc, n=30):
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n): fn()
        e1.record(s)
    s.synchronize(); ms=e0.elapsed_time(e1)/n
    print(f"{name:50s} {ms:.4f} ms {bpc*cells/ms/1e6/80:.1f} %", flush=True)
for segp in (1024, 256):
  for nt in (-1, 0):
    L.dlesm_set_tuning(b"util_segp", segp); L.dlesm_set_tuning(b"j5_nt_stores", nt)
    timed(f"segp={segp} nt={nt} pack", lambda: L.dlesm_pack_inner_f64(a.device_ptr,g.nx,g.ny,it.xstart,it.xstop,it.ystart,it.ystop,C.c_void_p(send.data_ptr()),cells,sp),16)
    timed(f"segp={segp} nt={nt} unpack", lambda: L.dlesm_unpack_gathered_f64(C.c_void_p(send.data_ptr()),cells,C.byref(pd._info),pd.subdomains,1,C.c_void_p(glob.data_ptr()),sp),16,n=10)
    timed(f"segp={segp} nt={nt} checksum async", lambda: L.dlesm_checksum_async_f64(a.device_ptr,g.nx,g.ny,it.xstart,it.xstop,it.ystart,it.ystop,C.c_void_p(res.data_ptr()),sp),8)
    timed(f"segp={segp} nt={nt} hash_init", lambda: D.psy.hash_init(b,7,stream=s),8)
    timed(f"segp={segp} nt={nt} fill box", lambda: L.dlesm_fill_f64(b.device_ptr,g.nx,g.ny,it.xstart,it.xstop,it.ystart,it.ystop,1.0,sp),8)
    if nt == -1:
        for (xa, xb) in ((1, g.nx), (1, g.nx - 1), (1, tile), (17, tile + 16), (2, tile), (1, 8192), (1, 15360)):
            timed(f"   fill columns {xa}..{xb} (ld {g.nx})", lambda: L.dlesm_fill_f64(b.device_ptr,g.nx,g.ny,xa,xb,it.ystart,it.ystop,1.0,sp), 8.0*(xb-xa+1)/tile)
