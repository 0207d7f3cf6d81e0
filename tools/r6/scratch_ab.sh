#!/bin/bash
# round 6, VERDICT r5 item 1: where did the in-flight gain go when every pipeline got its own launch scratch?
# One box, alternating: (a) own scratch, (b) ONE shared scratch (racy, timing only), (c) split-K off / capped at 2 / slab shrunk,
# (d) non-temporal slab stores (+ loads).  Then rocprofv3 kernel stats of (a) and (b) for the per-family times under three clips in flight.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_ab
O=gpurun_out/r6_ab
export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile --no-configs2 --steps 12 --warmup 3"
run() {  # tag, extra args
  tag=$1; shift
  timeout 300 python bench.py $B "$@" > $O/${tag}.json 2> $O/${tag}.err || echo "$tag failed rc=$?"
}
for r in 1 2; do
  run a_own_$r
  run b_shared_$r --shared-scratch --no-serial-leg
  run c_splitk0_$r --set-option split_k=0 --no-serial-leg
  run c_splitmax2_$r --set-option split_k_max=2 --no-serial-leg
  run c_slab88_$r --scratch-mb 88 --no-serial-leg
  run c_slab44_$r --scratch-mb 44 --no-serial-leg
  run d_nt1_$r --set-option splitk_nt=1 --no-serial-leg
  run d_nt2_$r --set-option splitk_nt=2 --no-serial-leg
done
for v in a b; do
  extra=""; [ $v = b ] && extra="--shared-scratch"
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o $v -- python bench.py --no-cpu-baseline --no-profile --no-configs2 --no-serial-leg --steps 9 --warmup 3 $extra > $O/prof_$v.log 2>&1
  python tools/prof_db_summary.py $O/prof_$v/${v}_results.db $O/prof_${v}_kernel_stats.csv $O/prof_${v}_gaps.json > /dev/null 2>&1
  rm -rf $O/prof_$v
done
python - <<'PY'
import glob, json, os
rows = []
for f in sorted(glob.glob("gpurun_out/r6_ab/*.json")):
    if "gaps" in f: continue
    try:
        d = json.load(open(f))
        rows.append((os.path.basename(f)[:-5], round(d["value"], 3), d.get("inflight_identity", {}).get("identical"),
                     round(d.get("one_clip_at_a_time", {}).get("value") or 0, 3)))
    except Exception as e:
        rows.append((os.path.basename(f), "failed", str(e)[:80], 0))
for r in rows: print(*r)
PY
