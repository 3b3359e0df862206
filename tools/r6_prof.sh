#!/bin/bash
# rocprofv3 kernel-trace statistics + separate PMC passes (never combined with other trace domains) of one command:
#     gpurun -- 'bash tools/r6_prof.sh <tag> python tools/run_cbp.py 64 5'
# writes gpurun_out/r6prof/<tag>_kernel_stats.csv and <tag>_pmc.csv
# PMC=0: kernel trace only.  KINC=<regex>: the PMC passes collect only kernels matching it (the kernel-trace pass always
# sees every kernel), e.g. KINC='hk::' inside a whole training step, where serialising MIOpen's kernels buys nothing.
set -u
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/r6prof
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
CMD="$@"
CMD=${CMD//tools\//$ROOT/tools/}
CMD=${CMD//bench.py/$ROOT/bench.py}
pass() {   # name, rocprofv3 options...
  local name=$1; shift
  rm -rf "$OUT/$TAG.$name"
  timeout ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace ${KINC:+--kernel-include-regex "$KINC"} "$@" --output-format csv -d "$OUT/$TAG.$name" -- $CMD > "$OUT/$TAG.$name.log" 2>&1 || echo "pass $name: rc=$?"
}
KINC_SAVE=${KINC:-}; KINC=
pass kt --stats
KINC=$KINC_SAVE
find "$OUT/$TAG.kt" -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} "$OUT/${TAG}_kernel_stats.csv"
if [ "${PMC:-1}" = "1" ]; then
  pass pmc_fetch --pmc FETCH_SIZE
  pass pmc_write --pmc WRITE_SIZE
  pass pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass pmc_lds --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  python $ROOT/tools/pmc_summary.py "$OUT/$TAG.pmc_fetch" "$OUT/$TAG.pmc_write" "$OUT/$TAG.pmc_mfma" "$OUT/$TAG.pmc_wait" "$OUT/$TAG.pmc_lds" --only hk:: > "$OUT/${TAG}_pmc.csv"
fi
rm -rf "$OUT/$TAG".kt "$OUT/$TAG".pmc_* 2>/dev/null
head -n 30 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-180
[ -f "$OUT/${TAG}_pmc.csv" ] && cat "$OUT/${TAG}_pmc.csv" | cut -c1-400
