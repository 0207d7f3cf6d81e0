import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hallo_amd import ops
dev = torch.device("cuda:0"); DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
def timeit(fn, nsets):
    for i in range(nsets): fn(i)
    torch.cuda.synchronize(); ts = []
    for _ in range(9):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            for i in range(nsets): fn(i)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / (4 * nsets))
    return round(sorted(ts)[4] * 1e3, 1)
for rows, Cd in [(16384, 640), (18432, 640), (4096, 1280), (4608, 1280), (1024, 1280)]:
    nsets = 12
    xs = [rnd(rows, Cd) for _ in range(nsets)]; ys = [torch.empty_like(xs[0]) for _ in range(nsets)]
    gm, bt = rnd(Cd), rnd(Cd); wq, wo, bo = rnd(Cd, Cd, sc=Cd**-0.5), rnd(Cd, Cd, sc=Cd**-0.5), rnd(Cd)
    kf, vf = rnd(1, 4, Cd), rnd(1, 4, Cd)
    sg, gg, bb, owp = ops.face_xattn_constants(wq, kf, vf, wo, gm, bt, 8, DT)
    rec = dict(rows=rows, C=Cd)
    for cap in (512, 384, 256, 192, 128, 64):
        ops.set_option("xattn_cap", cap)
        rec[f"us_cap{cap}"] = timeit(lambda i: ops.face_xattn(xs[i], sg, gg, bb, owp, bo, rows, 1e-5, out=ys[i]), nsets)
    ops.set_option("xattn_cap", 0)
    print(rec, flush=True)
