#!/bin/bash
# Round 3, final-state check:  gpurun --timeout 1500 -- 'bash tools/r3_final.sh'
# the whole GPU suite, smoke(), the default bench line (candidates, other models) - what the driver runs at round end.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3final
mkdir -p "$OUT"
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r3
mkdir -p "$HAWKEYE_MIOPEN_DIR"; cp -rn hawkeye_amd/miopen_db/* "$HAWKEYE_MIOPEN_DIR/" 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" ) > "$OUT/gpu_tests.log"
grep -E "passed|failed" "$OUT/gpu_tests.log" | tail -n 3; grep -E "^FAILED|^ERROR" "$OUT/gpu_tests.log" | head
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > "$OUT/smoke.log" 2>&1; tail -n 2 "$OUT/smoke.log"
( timeout 600 python bench.py 2> "$OUT/bench.err" ) > "$OUT/bench.json"; tail -n 3 "$OUT/bench.err"
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r3final/bench.json').read().strip().splitlines()[-1])
print({k: b[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus', 'dtype', 'config', 'vs_baseline')})
print('roofline', b.get('roofline')); print('cpu_baseline', b.get('cpu_baseline'))
for k in ('kernels',):
    print(k, [(x['kernel'], x['us'], x['frac']) for x in b.get(k, [])])
for r in b.get('other_models', []): print(r)
PY
