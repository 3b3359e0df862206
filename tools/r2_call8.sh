#!/bin/bash
# Round 2, GPU call 8: ROI backward (crop-only gather, compile-time windows) - tests, timing, cycle stamps
set -u
OUT=$PWD/gpurun_out/r2c8
mkdir -p "$OUT"
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "roi or cbp or compact or APCNN or apcnn" 2>&1 | tail -8 ) > "$OUT/gpu_tests.log"
timeout 120 python tools/roi_lab.py > "$OUT/roi_lab.json" 2> "$OUT/roi_lab.err"
python - > "$OUT/cand.json" 2> "$OUT/cand.err" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.roi_bwd)
c.guarded(c.cbp)
print(json.dumps(c.rows, indent=0))
PY
cat "$OUT/gpu_tests.log"; cat "$OUT/roi_lab.json"; tail -3 "$OUT/roi_lab.err" "$OUT/cand.err"; python -c "
import json
for r in json.load(open('$OUT/cand.json')): print({k: v for k, v in r.items() if k != 'flops'})
"
