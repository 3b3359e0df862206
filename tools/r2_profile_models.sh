#!/bin/bash
# Round 2: rocprofv3 kernel-trace statistics of the training step of the other BASELINE configs (channels_last)
#   gpurun --timeout 900 -- 'bash tools/r2_profile_models.sh'
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2prof_models
rm -rf "$OUT"; mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
cd /tmp; export TMPDIR=/tmp
for MB in MPN:64:200 CBCNN:64:200 APCNN:16:8142; do
  M=${MB%%:*}; R=${MB#*:}; BS=${R%%:*}; CL=${R##*:}
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step_$M" -- python $ROOT/bench.py --model $M --batch $BS --classes $CL --steps 3 --warmup 2 --no-cpu-baseline --no-kernels --no-candidates --no-other-models > "$OUT/step_$M.log" 2>&1 || echo "$M: rc=$?"
  find "$OUT/step_$M" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/r2_step_${M}_kernel_stats.csv"
  rm -rf "$OUT/step_$M"
  tail -n 1 "$OUT/step_$M.log" | cut -c1-200
done
ls -la "$OUT"
