#!/bin/bash
# Round 2, GPU call 2:  gpurun --timeout 1200 -- 'bash tools/r2_call2.sh'
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c2
mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
( timeout 300 python tools/ns_bench.py 5 2> "$OUT/ns_bench.err" ) > "$OUT/ns_bench.json"
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > "$OUT/gpu_tests.log"
( timeout 240 python tools/candidates.py 2> "$OUT/candidates.err" ) > "$OUT/candidates.json"
cd /tmp && export TMPDIR=/tmp
HK_NS_STREAMS=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ns2q" -o ns -- python "$ROOT/tools/ns_profile.py" 3 > "$OUT/prof_ns2q.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ns1q" -o ns -- python "$ROOT/tools/ns_profile.py" 3 > "$OUT/prof_ns1q.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" -size +3M -delete
grep -E "passed|failed" "$OUT/gpu_tests.log" | tail -3
grep -E "^FAILED|hip classifier" "$OUT/gpu_tests.log"
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r2c2/ns_bench.json')):
    print(r)
PY
