#!/bin/bash
# Round 2, GPU call 16: runner check, CBP backward with dc in LDS, diagonal K-block walk A/B
set -u
OUT=$PWD/gpurun_out/r2c16
mkdir -p "$OUT"
( timeout 120 python tools/run_pool_kernels.py 2 all ) > "$OUT/run_all.log" 2>&1; echo "run_all rc=$?"; tail -n 4 "$OUT/run_all.log"
( timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backward_128 or cbp or compact or refuse or roi or test_bcnn or osme" 2>&1 | tail -8 ) > "$OUT/gpu_tests.log"; cat "$OUT/gpu_tests.log"
timeout 150 python tools/bwd_ab.py > "$OUT/ab.json" 2> "$OUT/ab.err"; python -c "
import json
a = json.load(open('$OUT/ab.json')); print(json.dumps(a['us_per_round'])); print(a['median_us']); print(a['frac_of_157.3_TF'])"
python - > "$OUT/cand.json" 2> "$OUT/cand.err" <<'PY'
import sys, json
sys.path.insert(0, '.')
sys.argv = ['candidates']
import tools.candidates as c
c.guarded(c.cbp)
print(json.dumps(c.rows, indent=0))
PY
python -c "
import json
for r in json.load(open('$OUT/cand.json')):
    if 'bwd' in r['op'] or 'scatter' in r['variant']: print({k: v for k, v in r.items() if k != 'flops'})
"
