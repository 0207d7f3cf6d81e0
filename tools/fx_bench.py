import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hallo_amd import ops
dev = torch.device("cuda:0"); DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts)//2]
for rows, Cd in [(65536, 320), (16384, 640), (4096, 1280)]:
    x = rnd(rows, Cd); gm, bt = rnd(Cd), rnd(Cd); wq, wo, bo = rnd(Cd, Cd, sc=Cd**-0.5), rnd(Cd, Cd, sc=Cd**-0.5), rnd(Cd)
    kf, vf = rnd(1, 4, Cd), rnd(1, 4, Cd)
    sg, gg, bb, owp = ops.face_xattn_constants(wq, kf, vf, wo, gm, bt, 8, DT)
    ms = timeit(lambda: ops.face_xattn(x, sg, gg, bb, owp, bo, rows, 1e-5))
    print(f"face_xattn rows={rows} C={Cd}: {ms*1000:.1f} us  {2*2*rows*Cd/ms/1e6:.0f} GB/s", flush=True)
    # unfused chain
    def chain():
        nh = ops.layernorm(x, gm, bt, 1e-5)
        q = ops.gemm(nh, wq, None, alpha=0.2)
        a = ops.attention(q.view(16, rows//16, Cd), kf.expand(16,4,Cd).contiguous(), vf.expand(16,4,Cd).contiguous(), 8, q_prescaled=True)
        return ops.gemm(a.view(rows, Cd), wo, bo, residual=x)
    ms2 = timeit(chain)
    print(f"   unfused chain: {ms2*1000:.1f} us", flush=True)
