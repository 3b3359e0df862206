#!/bin/bash
# Round 3 profiles:  gpurun --timeout 1300 -- 'bash tools/r3_profile.sh'
# rocprofv3 kernel-trace statistics and separate PMC passes (never combined with other trace domains) of the head
# kernels at the BASELINE shapes (tools/run_pool_kernels.py), the Newton-Schulz launch timeline, and kernel-trace
# statistics of the four training steps.  Summaries land in gpurun_out/r3prof/ (copied to profiles/ by hand).
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3prof
rm -rf "$OUT"; mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r3
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
cd /tmp; export TMPDIR=/tmp
RUN="python $ROOT/tools/run_pool_kernels.py 3 all"
( timeout 150 $RUN ) > "$OUT/run_all.log" 2>&1 || { echo "run_pool_kernels failed - nothing profiled"; tail -n 5 "$OUT/run_all.log"; exit 1; }
pass() {   # name, rocprofv3 options...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace "$@" --output-format csv -d "$OUT/$name" -- $RUN > "$OUT/$name.log" 2>&1 || echo "pass $name: rc=$?"
}
pass kt --stats
find "$OUT/kt" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r3_head_kernel_stats.csv"
if [ "${PMC:-1}" = "1" ]; then
  pass pmc_fetch --pmc FETCH_SIZE
  pass pmc_write --pmc WRITE_SIZE
  pass pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass pmc_lds --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  python $ROOT/tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_wait" "$OUT/pmc_lds" --only hk:: > "$OUT/r3_pool_kernels_pmc.csv"
fi
# Newton-Schulz chain: per-kernel statistics and the two-queue launch timeline of the last repetition
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ns" -- python $ROOT/tools/ns_profile.py 4 > "$OUT/ns.log" 2>&1 || echo "ns: rc=$?"
find "$OUT/ns" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r3_ns_kernel_stats.csv"
python $ROOT/tools/ns_timeline.py "$OUT/ns" > "$OUT/r3_ns_launch_timeline.csv" 2> "$OUT/ns_timeline.err" || tail -n 3 "$OUT/ns_timeline.err"
# the training steps (channels_last), kernel-trace statistics
if [ "${STEPS:-1}" = "1" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_BCNN" -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-candidates --no-other-models > "$OUT/step_BCNN.log" 2>&1 || echo "BCNN: rc=$?"
  find "$OUT/step_BCNN" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r3_step_BCNN_kernel_stats.csv"
  tail -n 1 "$OUT/step_BCNN.log" | cut -c1-300
  for MB in MPN:64:200 CBCNN:64:200 APCNN:16:8142; do
    M=${MB%%:*}; R=${MB#*:}; BS=${R%%:*}; CL=${R##*:}
    timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_$M" -- python $ROOT/bench.py --model $M --batch $BS --classes $CL --steps 3 --warmup 2 --no-cpu-baseline --no-kernels --no-candidates --no-other-models > "$OUT/step_$M.log" 2>&1 || echo "$M: rc=$?"
    find "$OUT/step_$M" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r3_step_${M}_kernel_stats.csv"
    tail -n 1 "$OUT/step_$M.log" | cut -c1-200
  done
fi
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/kt "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_mfma "$OUT"/pmc_wait "$OUT"/pmc_lds "$OUT"/ns "$OUT"/step_BCNN "$OUT"/step_MPN "$OUT"/step_CBCNN "$OUT"/step_APCNN 2>/dev/null
ls -la "$OUT"; head -n 45 "$OUT/r3_head_kernel_stats.csv" | cut -c1-170; cat "$OUT/r3_ns_launch_timeline.csv" | head -n 50
