#!/bin/bash
# Round 2, GPU call 4:  gpurun --timeout 1200 -- 'bash tools/r2_call4.sh'
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c4
mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
( timeout 120 python tools/bwd_lab.py 2> "$OUT/bwd_lab.err" ) > "$OUT/bwd_lab.json"
( timeout 200 python tools/ns_bench.py 3 2> "$OUT/ns_bench.err" ) > "$OUT/ns_bench.json"
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > "$OUT/gpu_tests.log"
( timeout 240 python tools/candidates.py 2> "$OUT/candidates.err" ) > "$OUT/candidates.json"
( timeout 300 python bench.py --steps 20 --warmup 5 --kernels-first --no-candidates --no-other-models 2> "$OUT/bench_kf.err" ) > "$OUT/bench_kf.json"
grep -E "passed|failed" "$OUT/gpu_tests.log" | tail -3
grep -E "^FAILED" "$OUT/gpu_tests.log"
cat "$OUT/bwd_lab.json"; tail -3 "$OUT/bwd_lab.err"
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r2c4/ns_bench.json')):
    print({k: v for k, v in r.items() if not k.endswith('_all')})
for r in json.load(open('gpurun_out/r2c4/candidates.json')):
    if 'roi' in r.get('op', '') or 'error' in r:
        print(r)
b = json.load(open('gpurun_out/r2c4/bench_kf.json'))
print(b['value'], b['ms_per_step'])
for k in ('kernels_before', 'kernels'):
    print(k, [(x['kernel'], x['us'], x['frac']) for x in b[k]])
PY
