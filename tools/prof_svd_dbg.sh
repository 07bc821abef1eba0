#!/bin/bash
# timing breakdown of jacb_eigen_x: ND4HIP_JAC_XDBG = 1 (exit after the load), 2 (after the pre-check), 3 (no rounds), 0 (full)
# usage (GPU box): bash tools/prof_svd_dbg.sh <outdir> [xkernel variants...]
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for xk in ${@:-8}; do for d in ${XDBGS:-1 2 3 0}; do
  ND4HIP_JAC_XKERNEL=$xk ND4HIP_JAC_XDBG=$d ND4HIP_SVD_MAXSWEEPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $out/x${xk}dbg$d -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py svd --reps 1 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('$out/x${xk}dbg$d/p_kernel_stats.csv')):
    n = r['Name']
    if 'jacb' in n: print('xk=$xk dbg=$d', n.replace('(anonymous namespace)::','').replace('void ','')[:18], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', int(r['MinNs'])/1e3, 'max', int(r['MaxNs'])/1e3)
PY
done; done
