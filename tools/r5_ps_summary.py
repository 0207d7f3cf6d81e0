import glob, json, os
for f in sorted(glob.glob("gpurun_out/r5_bench_ps[01]_*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(f)); k = d.get("kernels", {})
        print(os.path.basename(f), round(d["value"], 3), d.get("inflight_identity", {}).get("identical"),
              {n: (k[n]["ms"], k[n]["launches"]) for n in ("gemm", "row_stats", "face_xattn") if n in k}, d.get("kernel_ms_per_clip"))
    except Exception as e:
        print(f, "failed", e)
