#!/bin/bash
# Round 2, GPU call 18: cyclic Gram walk (cov fwd C=256, CBP B=16), AP-CNN train-mode pin, full suite
set -u
OUT=$PWD/gpurun_out/r2c18
mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$PWD/gpurun_out/miopen_r2
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > "$OUT/gpu_tests.log"
grep -E "passed|failed" "$OUT/gpu_tests.log" | tail -n 3; grep -E "^FAILED|^ERROR" "$OUT/gpu_tests.log" | head
grep -B2 -A12 "def test_apcnn_train_mode" "$OUT/gpu_tests.log" | grep -E "assert|Error" | head
( timeout 300 python tools/model_rows.py --kernels-only 2> "$OUT/rows.err" ) > "$OUT/rows.json" || true
python - <<'PY'
import json
try:
    rows = json.load(open('gpurun_out/r2c18/rows.json'))
    for r in rows:
        if r.get('model') in ('MPN', 'CBCNN') and 'kernel' in r: print(r)
except Exception as e:
    print('rows:', e)
PY
tail -n 3 "$OUT/rows.err"
