#!/bin/bash
# Round 2, GPU call 3:  gpurun --timeout 1200 -- 'bash tools/r2_call3.sh'
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c3
mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
( timeout 300 python tools/ns_bench.py 5 2> "$OUT/ns_bench.err" ) > "$OUT/ns_bench.json"
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > "$OUT/gpu_tests.log"
( timeout 240 python tools/candidates.py 2> "$OUT/candidates.err" ) > "$OUT/candidates.json"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_pool" -o pool -- python "$ROOT/tools/run_pool_kernels.py" 5 > "$OUT/prof_pool.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_pool_fetch" -o pool -- python "$ROOT/tools/run_pool_kernels.py" 3 > "$OUT/pmc_pool_fetch.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_pool_sq" -o pool -- python "$ROOT/tools/run_pool_kernels.py" 3 > "$OUT/pmc_pool_sq.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ns1q" -o ns -- python "$ROOT/tools/ns_profile.py" 3 > "$OUT/prof_ns1q.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" -size +3M -delete
grep -E "passed|failed" "$OUT/gpu_tests.log" | tail -3
grep -E "^FAILED|hip classifier" "$OUT/gpu_tests.log"
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r2c3/ns_bench.json')):
    print(r)
PY
python - <<'PY'
import json
try:
    for r in json.load(open('gpurun_out/r2c3/candidates.json')):
        if any(k in r.get('op','') for k in ('bwd', 'linear fwd bcnn', 'cbp fwd')):
            print({k: v for k, v in r.items() if k not in ('note', 'gbs', 'bound', 'unit')})
except Exception as e:
    print('candidates unreadable', e)
PY
