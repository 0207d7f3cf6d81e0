#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
B="--no-cpu-baseline --no-serial-leg --no-configs2 --steps 3 --warmup 3 --shape-breakdown"
for ps in 0 1; do
  timeout 300 python bench.py $B --set-option producer_stats=$ps > $O/r5_bd_ps$ps.json 2> $O/r5_bd_ps$ps.err
  mv $O/shape_breakdown.json $O/r5_shape_breakdown_ps$ps.json
done
python - <<'PY'
import json
a={ (r["op"],r["shapes"]):r for r in json.load(open("gpurun_out/r5_shape_breakdown_ps0.json"))}
b={ (r["op"],r["shapes"]):r for r in json.load(open("gpurun_out/r5_shape_breakdown_ps1.json"))}
import re
def key(k): return (k[0], re.sub(r"\('sym'.*", "", k[1]))
A={}; B={}
for k,v in a.items(): A.setdefault(key(k),[0,0]); A[key(k)][0]+=v["ms"]; A[key(k)][1]+=v["launches"]
for k,v in b.items(): B.setdefault(key(k),[0,0]); B[key(k)][0]+=v["ms"]; B[key(k)][1]+=v["launches"]
rows=[]
for k in set(A)|set(B):
    x=A.get(k,[0,0]); y=B.get(k,[0,0])
    rows.append((y[0]-x[0],k,x,y))
for d,k,x,y in sorted(rows,key=lambda r:-abs(r[0]))[:28]:
    print(round(d,2),k[0],k[1][:110],"| ps0",round(x[0],2),x[1],"| ps1",round(y[0],2),y[1])
PY
