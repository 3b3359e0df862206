#!/bin/bash
# Per-dispatch kernel trace of one short BCNN run: which kernel follows each of MIOpen's SubTensorOpWithScalar1d zero-fills?
#   gpurun -- 'bash tools/probe/trace_seq.sh'
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/trseq
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trseq -- python $ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels --no-other-models > /tmp/trseq.log 2>&1
f=$(find /tmp/trseq -name "*kernel_trace.csv" | head -n 1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
last = rows[int(n * 0.75):]                      # the last (timed) step(s)
out = {}
for i, r in enumerate(last[:-1]):
    if 'SubTensorOpWithScalar1d' in r['Kernel_Name']:
        nxt = last[i + 1]['Kernel_Name'][:60]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        k = (nxt, r.get('Grid_Size_X', r.get('Grid_Size', '?')))
        out.setdefault(nxt, []).append(d)
for k, v in sorted(out.items(), key=lambda kv: -sum(kv[1])):
    print(f'{len(v):4d} fills, {sum(v):9.1f} us in all, longest {max(v):7.1f} us  -> followed by {k}')
PY
