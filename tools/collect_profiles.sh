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
ks svd2048 python3 $R/tools/prof_ops.py svd
ks lu_qr_chol_2048 python3 $R/tools/prof_ops.py lu qr chol
ks qr_panel_batched python3 $R/tools/prof_panel.py 256 2048 2048 512 1024 1024
ks lu_qr_batch1024x512 python3 $R/tools/prof_batch.py 1024 lu qr
ks svd_batch256x512 python3 $R/tools/prof_batch.py 256
ks svd_small python3 $R/tools/prof_small.py
ks hess2048 python3 $R/tools/prof_hess.py 2048 hess
ks bidiag2048 python3 $R/tools/prof_hess.py 2048 bidiag
ks qr4096 python3 $R/tools/prof_ops.py qr --n 4096 --reps 2
ks lu4096 python3 $R/tools/prof_ops.py lu --n 4096 --reps 2
pmc gemm_fetch FETCH_SIZE python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc gemm_write WRITE_SIZE python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc gemm_busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 $R/bench.py --steps 5 --warmup 1 --no-ops --no-cpu-baseline
pmc panel_fetch FETCH_SIZE python3 $R/tools/prof_panel.py 256 2048 2048 512
pmc panel_write WRITE_SIZE python3 $R/tools/prof_panel.py 256 2048 2048 512
pmc qrlu2048_fetch FETCH_SIZE python3 $R/tools/prof_ops.py qr lu --reps 1
pmc qrlu2048_write WRITE_SIZE python3 $R/tools/prof_ops.py qr lu --reps 1
pmc small_fetch FETCH_SIZE python3 $R/tools/prof_small.py
pmc small_write WRITE_SIZE python3 $R/tools/prof_small.py
pmc hess_fetch FETCH_SIZE python3 $R/tools/prof_hess.py 2048 hess bidiag
pmc hess_write WRITE_SIZE python3 $R/tools/prof_hess.py 2048 hess bidiag
pmc svd_fetch FETCH_SIZE python3 $R/tools/prof_ops.py svd --reps 1
pmc svd_write WRITE_SIZE python3 $R/tools/prof_ops.py svd --reps 1
# summaries (tools/pmc_summary.py: FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes)
S() { python3 $R/tools/pmc_summary.py "$@" > /dev/null 2>$out/pmc_summary.err || echo "pmc_summary failed: $1"; }
S $out/gemm4096_pmc.json dgemm_kernel $out/gemm_fetch/p_counter_collection.csv $out/gemm_write/p_counter_collection.csv $out/gemm_busy/p_counter_collection.csv
S $out/qr_panel_batched_pmc.json "qrb_panel<4, 8;qrb_panel<4, 2" $out/panel_fetch/p_counter_collection.csv $out/panel_write/p_counter_collection.csv
S $out/qr_lu_2048_pmc.json "qrh_bc<4,qrh_bc<2,qrh_bc<1,lu_panel_row_la<4,lu_panel_row_la<2,lu_panel_row_la<1,lu_narrow_fused" $out/qrlu2048_fetch/p_counter_collection.csv $out/qrlu2048_write/p_counter_collection.csv
S $out/svd_small_pmc.json "jac_small<32" $out/small_fetch/p_counter_collection.csv $out/small_write/p_counter_collection.csv
S $out/hess_bidiag_2048_pmc.json "hessp<8;bdp<8" $out/hess_fetch/p_counter_collection.csv $out/hess_write/p_counter_collection.csv
S $out/svd2048_pmc.json "jacb_eigen_pu,jacb_apply_w,jacb_gram2" $out/svd_fetch/p_counter_collection.csv $out/svd_write/p_counter_collection.csv
# the in-kernel phase stamps: the batched QR panel (workgroup 0), the one-launch QR panels of one matrix; the ceilings of the panel's
# data movement (copy-only modes) and of a plain device copy
ND4HIP_QRB_STAMPS=1 python3 $R/tools/panel_stamps.py 2>&1 | grep 'qrb stamps' > $out/qr_panel_batched_phase_stamps.txt
ND4HIP_QR_STAMPS=1 python3 $R/tools/prof_ops.py qr --n 2048 --reps 1 2>&1 | grep 'qrh stamp' > $out/qr2048_panel_phase_stamps.txt
ND4HIP_HESSP_STAMPS=1 ND4HIP_BDP_STAMPS=1 python3 $R/tools/time_hess.py 2048 2>&1 | grep -E '^hessp|^bdp' | sort -u > $out/hess_bidiag_phase_stamps.txt
( echo "== thread-per-row copy (ND4HIP_QRB_COPY_ONLY=1)"; ND4HIP_QRB_COPY_ONLY=1 python3 $R/tools/time_panel.py 2>/dev/null; echo "== whole-line copy (=2)"; ND4HIP_QRB_COPY_ONLY=2 python3 $R/tools/time_panel.py 2>/dev/null; echo "== plain device copy"; python3 $R/tools/copy_bw.py ) > $out/qr_panel_copy_ceiling.txt 2>&1
python3 $R/tools/time_batch.py 1024 lu qr > $out/lu_qr_batch_times.txt 2>&1
python3 $R/bench.py > $out/bench_default.json 2> $out/bench_default.err
rm -rf $out/*/p_kernel_trace.csv $out/*/p_agent_info.csv $out/*/p_domain_stats.csv
ls $out
