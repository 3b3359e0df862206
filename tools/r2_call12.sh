#!/bin/bash
# Round 2, GPU call 12: Gram backward staged by LDS-DMA - tests and timing
set -u
OUT=$PWD/gpurun_out/r2c12
mkdir -p "$OUT"
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backward_128 or signed_sqrt or bcnn or bilinear or BCNN" 2>&1 | tail -8 ) > "$OUT/gpu_tests.log"
python - > "$OUT/cand.json" 2> "$OUT/cand.err" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.bwd_variants)
print(json.dumps(c.rows, indent=0))
PY
cat "$OUT/gpu_tests.log"; tail -n 3 "$OUT/cand.err"; python -c "
import json
for r in json.load(open('$OUT/cand.json')): print({k: v for k, v in r.items() if k != 'flops'})
"
