#!/bin/bash
# Second gpurun call of the next round, AFTER the winners of tools/round2_first_call.sh were made the defaults:
#   gpurun --timeout 1200 -- 'bash tools/round2_second_call.sh'
# full GPU suite + smoke + bench, the rocprofv3 kernel-trace of the bench, and the PMC passes (separate passes, never
# combined with other trace domains) for the pooling-head kernels.  Everything lands in gpurun_out/r2_second/.
set -u
OUT=$PWD/gpurun_out/r2_second
mkdir -p "$OUT"
ROOT=$PWD
( timeout 400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > "$OUT/pytest_gpu.txt"
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > "$OUT/smoke.txt"
( timeout 420 python bench.py 2> "$OUT/bench.err" | tail -1 ) > "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-candidates > /dev/null 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        python "$ROOT/tools/run_pool_kernels.py" 3 ns > /dev/null 2>&1
done
cd "$ROOT"
cat "$OUT/pytest_gpu.txt" | tail -3; cat "$OUT/smoke.txt"; cut -c1-600 "$OUT/bench.json"
