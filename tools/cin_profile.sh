#!/bin/bash
# CIN kernels (hk_cin_* at configs/CIN.yaml's shape, tools/cin_rows.py): kernel-trace statistics and separate PMC passes.
#   gpurun --timeout 900 -- 'bash tools/cin_profile.sh'        -> gpurun_out/cinprof/cin_kernel_stats.csv, cin_pmc.csv
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/cinprof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
RUN="python $ROOT/tools/cin_rows.py"
pass() {
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace "$@" --output-format csv -d "$OUT/$name" -- $RUN > "$OUT/$name.log" 2>&1 || echo "pass $name: rc=$?"
}
pass kt --stats
find "$OUT/kt" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/cin_kernel_stats.csv"
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python $ROOT/tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_wait" "$OUT/pmc_tcc" --only hk:: > "$OUT/cin_pmc.csv"
rm -rf "$OUT"/kt "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_wait "$OUT"/pmc_tcc
