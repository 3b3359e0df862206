#!/bin/bash
# Round 2, GPU call 6: ROI backward with register weights, CBP finishing stage + two queues
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c6
mkdir -p "$OUT"
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "roi or cbp or compact or CBCNN or APCNN or apcnn" 2>&1 | tail -15 ) > "$OUT/gpu_tests.log"
python - > "$OUT/cand.json" 2> "$OUT/cand.err" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.roi_bwd)
c.guarded(c.cbp)
print(json.dumps(c.rows, indent=0))
PY
cat "$OUT/gpu_tests.log"; tail -3 "$OUT/cand.err"; python -c "
import json
for r in json.load(open('$OUT/cand.json')): print({k: v for k, v in r.items() if k != 'flops'})
"
