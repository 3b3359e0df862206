#!/bin/bash
# Round 2, GPU call 19: eight-wave 64-row backward (covariance C=256) - tests and timing
set -u
OUT=$PWD/gpurun_out/r2c19
mkdir -p "$OUT"
( timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backward_128 or cov or mpn or MPN" 2>&1 | tail -5 ) > "$OUT/gpu_tests.log"; cat "$OUT/gpu_tests.log"
python - > "$OUT/cand.json" 2> "$OUT/cand.err" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.bwd_variants)
c.guarded(c.bwd_variants)
print(json.dumps(c.rows, indent=0))
PY
tail -n 3 "$OUT/cand.err"; python -c "
import json
for r in json.load(open('$OUT/cand.json')): print({k: v for k, v in r.items() if k not in ('flops', 'note', 'gbs')})
"
