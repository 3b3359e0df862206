#!/bin/bash
# Round 2 profiles:  gpurun --timeout 1100 -- 'bash tools/r2_profile.sh'
# rocprofv3 kernel-trace statistics and separate PMC passes (never combined with other trace domains) of the head
# kernels at the BASELINE shapes (tools/run_pool_kernels.py), and kernel-trace statistics of the BCNN training step.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2prof
rm -rf "$OUT"; mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
cd /tmp; export TMPDIR=/tmp
RUN="python $ROOT/tools/run_pool_kernels.py 3 all"
( timeout 120 $RUN ) > "$OUT/run_all.log" 2>&1 || { echo "run_pool_kernels failed - nothing profiled"; tail -n 5 "$OUT/run_all.log"; exit 1; }
pass() {   # name, rocprofv3 options...
  local name=$1; shift
  timeout 170 rocprofv3 --kernel-trace "$@" --output-format csv -d "$OUT/$name" -- $RUN > "$OUT/$name.log" 2>&1 || echo "pass $name: rc=$?"
}
pass kt --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass pmc_lds --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python $ROOT/tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_wait" "$OUT/pmc_lds" --only hk:: > "$OUT/r2_pool_kernels_pmc.csv"
find "$OUT/kt" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r2_head_kernel_stats.csv"
# the BCNN training step (channels_last), kernel-trace statistics
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_BCNN" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-candidates --no-other-models > "$OUT/step_BCNN.log" 2>&1
find "$OUT/step_BCNN" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r2_step_BCNN_kernel_stats.csv"
tail -n 1 "$OUT/step_BCNN.log" | cut -c1-400
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/kt "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_mfma "$OUT"/pmc_wait "$OUT"/pmc_lds "$OUT"/step_BCNN 2>/dev/null
ls -la "$OUT"; grep -E "bcnn_bwd|gram_panel|nsmm|roi_crop" "$OUT/r2_pool_kernels_pmc.csv" | head -n 60; head -n 40 "$OUT/r2_head_kernel_stats.csv" | cut -c1-200
