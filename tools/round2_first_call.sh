#!/bin/bash
# First gpurun call of the next round, in one go (GPU minutes are the scarce resource):
#   gpurun --timeout 900 -- 'bash tools/round2_first_call.sh'
# 1. the candidate suites on the real device (first GPU run of the rows written after round 1's budget was spent)
# 2. timings of every opt-in variant next to what it would replace (tools/candidates.py)
# 3. rocprofv3 kernel-trace of the same (per-kernel durations: NS tile variants, split-K linear, CIN, n-pairs)
# Everything lands in gpurun_out/r2_first/ ; copy what should be judged into profiles/.
set -u
OUT=$PWD/gpurun_out/r2_first
mkdir -p "$OUT"
ROOT=$PWD
( timeout 300 python -m pytest tests/test_gpu_zz_candidates.py -q -m gpu 2>&1 | tail -15 ) > "$OUT/pytest_candidates.txt"
( timeout 200 python tools/candidates.py --step 2> "$OUT/candidates.err" ) > "$OUT/candidates.json"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o cand -- \
    python "$ROOT/tools/candidates.py" > /dev/null 2>&1
cd "$ROOT"
head -40 "$OUT"/prof/*kernel_stats.csv 2>/dev/null | cut -c1-200 > "$OUT/kernel_stats_head.txt"
tail -5 "$OUT/pytest_candidates.txt"
OUT="$OUT" python - <<'PY'
import json, os
p = os.path.join(os.environ.get('OUT', 'gpurun_out/r2_first'), 'candidates.json')
try:
    for r in json.load(open(p)):
        print({k: r[k] for k in r if k in ('op', 'variant', 'us', 'tflops', 'gbs', 'ms_per_step', 'images_per_sec', 'error',
                                           'rel_vs_default', 'rel_err_vs_torch')})
except Exception as e:
    print('no candidates.json:', e)
PY
