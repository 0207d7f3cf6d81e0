"""Which stored-oracle fingerprints (tests/golden/full_size_golden.npz) does THIS host reproduce?  Inputs first (seconds), then
the synthetic weights (builds the fp32 oracle nets: ~1-2 min)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_full_size_gpu as T
g = T._golden()["meta"]
out = {"cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"), "threads": torch.get_num_threads(),
       "cpu_capability": torch.backends.cpu.get_cpu_capability()}
for case in ("512x512x16f-cfg", "768x768x24f"):
    B, Fr, h = T.CASES[case]
    d = T._unet_inputs(B, Fr, h)
    f = T.fingerprint(T._unet_input_list(d, B, h))
    m = g[f"unet3d/B{B}_F{Fr}_h{h}"]["inputs"]
    out[case] = {"here": f, "stored": m, "exact": f["bits"] == m["bits"], "same_data": T.same_data(m, f)}
d, args, lat, flat, geo = T._pipe10_inputs()
f = T.fingerprint(flat)
out["pipeline10"] = {"here": f, "stored": g["pipeline10"]["inputs"], "exact": f["bits"] == g["pipeline10"]["inputs"]["bits"],
                     "same_data": T.same_data(g["pipeline10"]["inputs"], f)}
t0 = time.time()
for names, key in ((("denoising_unet", "reference_unet"), "unet3d/B2_F16_h64"), (T.PIPE_NETS, "pipeline10")):
    f = T._weights_fp(names)
    m = g[key]["weights"]
    out["weights:" + key] = {"here": f, "stored": m, "exact": f["bits"] == m["bits"], "same_data": T.same_data(m, f)}
out["weights_seconds"] = round(time.time() - t0, 1)
print(json.dumps(out, indent=1))
