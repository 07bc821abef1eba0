#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box: bash tools/collect_profiles.sh <outdir under gpurun_out>
# kernel-trace/--stats runs and --pmc runs are separate invocations (the pool refuses them combined with other trace domains),
# the program comes directly after `--`.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ks() { rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1 -o p -- "${@:2}" > $out/$1.stdout 2> $out/$1.stderr; cp $out/$1/p_kernel_stats.csv $out/$1_kernel_stats.csv 2>/dev/null; echo "done $1"; }
pmc() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $out/$1 -o p -- "${@:3}" > /dev/null 2>&1; echo "done pmc $1"; }
ks bench_matmul4096 python3 $R/bench.py --steps 20 --warmup 3 --no-ops --no-cpu-baseline
ks bench_default python3 $R/bench.py
ks svd2048 python3 $R/tools/prof_ops.py svd
ks lu_qr_chol_2048 python3 $R/tools/prof_ops.py lu qr chol
ks svd_batch256x512 python3 $R/tools/prof_batch.py 256
ks hess2048 python3 $R/tools/prof_hess.py 2048 hess
ks bidiag2048 python3 $R/tools/prof_hess.py 2048 bidiag
ks lu_solve2048 python3 $R/tools/prof_ops.py lusolve
ks qr4096 python3 $R/tools/prof_ops.py qr --n 4096 --reps 2
ks lu_qr_8192 python3 $R/tools/prof_ops.py lu qr --n 8192 --reps 1
ks lu4096 python3 $R/tools/prof_ops.py lu --n 4096 --reps 2
pmc gemm_fetch FETCH_SIZE python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc gemm_write WRITE_SIZE python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc gemm_busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc lu8192_fetch FETCH_SIZE python3 $R/tools/prof_ops.py lu --n 8192 --reps 1
pmc lu8192_write WRITE_SIZE python3 $R/tools/prof_ops.py lu --n 8192 --reps 1
pmc svd_fetch FETCH_SIZE python3 $R/tools/prof_ops.py svd --reps 1
pmc svd_write WRITE_SIZE python3 $R/tools/prof_ops.py svd --reps 1
# the in-kernel exchange (tools/xwg_lat.hip) and the phase stamps of the one-launch QR panels / the multi-workgroup LU panels
hipcc --offload-arch=gfx950 -O3 $R/tools/xwg_lat.hip -o $out/xwg_lat && timeout -k 10 120 $out/xwg_lat > $out/xwg_lat.txt 2>&1; rm -f $out/xwg_lat
ND4HIP_QR_STAMPS=1 python3 $R/tools/prof_ops.py qr --n 2048 --reps 1 2>&1 | grep 'qrh stamp' > $out/qr2048_panel_phase_stamps.txt
ND4HIP_LU_MW_R=2 ND4HIP_LU_STAMPS=1 ND4HIP_LU_NO_LOOKAHEAD=1 python3 $R/tools/prof_ops.py lu --n 4096 --reps 1 2>&1 | grep 'mw stamps' > $out/lu4096_mw_panel_phase_stamps.txt
ls $out
