import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hallo_amd import ops
dev = torch.device("cuda:0"); DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
def ev(fn, it=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
M, K = 65536, 320
x = rnd(M, K) + 0.3
gamma, beta = rnd(K, sc=0.1) + 1.0, rnd(K, sc=0.1)
for N, geglu in ((960, False), (1280, True)):
    w = rnd((2 * N if geglu else N), K, sc=K ** -0.5); b = rnd(2 * N if geglu else N)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    run = lambda: ops.gemm(x, wf, bf, geglu=geglu, ln_colsum=cs, ln_eps=1e-5)
    for dbg in (6, 14, 6, 14):
        ops.set_option("gemm_rs_dbg", dbg)
        print(N, geglu, "dbg", dbg, round(ev(run), 1), "us")
ops.set_option("gemm_rs_dbg", 0)
