#!/bin/bash
# Round 2, GPU call 1:  gpurun --timeout 1100 -- 'bash tools/r2_call1.sh'
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2c1
mkdir -p "$OUT" "$ROOT/gpurun_out/miopen_r2"
cp -rn hawkeye_amd/miopen_db/* "$ROOT/gpurun_out/miopen_r2/" 2>/dev/null || true
export HAWKEYE_MIOPEN_DIR=$ROOT/gpurun_out/miopen_r2
rocminfo | grep -E "Marketing|gfx" | head -4 > "$OUT/device.txt"
# 1. variant timings (NS grouped kernel vs tile widths, backward v0/v4, ROI, linear slabs, CBP binning, CIN)
( timeout 240 python tools/candidates.py 2> "$OUT/candidates.err" ) > "$OUT/candidates.json"
# 2. the whole GPU suite (no -x: every failure is wanted), prints kept
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -120 ) > "$OUT/gpu_tests.log"
# 3. other BASELINE configs
( timeout 500 python tools/model_rows.py 2> "$OUT/model_rows.err" ) > "$OUT/model_rows.json"
# 4. RCCL path on one GPU + bucket timeline
( timeout 200 python bench.py --force-pg --steps 3 --warmup 2 --no-cpu-baseline --no-kernels --no-candidates 2> "$OUT/bench_forcepg.err" ) > "$OUT/bench_forcepg.json"
# 5. rocprofv3 of the MPN head: kernel trace, then counters in their own passes
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ns" -o ns -- python "$ROOT/tools/ns_profile.py" 3 > "$OUT/prof_ns.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o ns -- python "$ROOT/tools/ns_profile.py" 2 > "$OUT/pmc_sq.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o ns -- python "$ROOT/tools/ns_profile.py" 2 > "$OUT/pmc_fetch.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o ns -- python "$ROOT/tools/ns_profile.py" 2 > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" -size +3M -delete        # traces can be large; the stats files are what is wanted
du -sh "$OUT" gpurun_out/miopen_r2
tail -5 "$OUT/gpu_tests.log"
python - <<'PY'
import json
for f in ('gpurun_out/r2c1/candidates.json', 'gpurun_out/r2c1/model_rows.json'):
    try:
        for r in json.load(open(f)):
            print({k: v for k, v in r.items() if k not in ('note', 'gbs', 'tflops', 'bound', 'unit')})
    except Exception as e:
        print(f, 'unreadable:', e)
PY
cat "$OUT/bench_forcepg.json" | cut -c1-1500
