cd /tmp && export TMPDIR=/tmp
for d in 0 1 2; do
ND4HIP_G2DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2s/d$d -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py svd --reps 1 > /dev/null 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r2s/d$d/p_kernel_stats.csv")))[:4]:
    if "gram" in r["Name"]: print("dbg $d", r["Name"].replace("(anonymous namespace)::","")[:14], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "min", int(r["MinNs"])/1e3)
PY
done
