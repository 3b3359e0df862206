#!/bin/bash
# Round 2, GPU call 5: Gram-backward lab
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c5
mkdir -p "$OUT"
( timeout 120 python tools/bwd_lab.py 2> "$OUT/bwd_lab.err" ) > "$OUT/bwd_lab.json"
( timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backward_128 or signed_sqrt or bcnn or cov or cbp" 2>&1 | tail -5 ) > "$OUT/gpu_tests_bwd.log"
python - > "$OUT/bwd_time.json" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.bwd_variants)
print(json.dumps(c.rows))
PY
cat "$OUT/bwd_lab.json"; tail -2 "$OUT/bwd_lab.err"; cat "$OUT/gpu_tests_bwd.log"; cat "$OUT/bwd_time.json" | cut -c1-1500
