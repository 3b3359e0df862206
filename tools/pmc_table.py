"""Markdown table from one rocprofv3 kernel-stats CSV + the PMC summary of the same command (tools/r6_prof.sh):
    python tools/pmc_table.py profiles/r6_step_MPN_kernel_stats.csv profiles/r6_step_MPN_pmc.csv
per hk:: kernel: launches, mean us, HBM bytes per launch from the counters (FETCH_SIZE doubled: gfx950 reports half of the
bytes of 16-B-per-lane streaming reads, MI355X_MICROARCH.md section HBM; KiB), matrix-pipe busy cycles per SIMD and what
share of the launch that is at 2.4 GHz, LDS bank-conflict cycles / LDS-active cycles, parked wave time (SQ_WAIT_ANY /
SQ_WAVE_CYCLES)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace('void ', '')
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*$', '', name).strip()
    return name.replace('hk::', '')


stats = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'hk::' in r['Name']:
        stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
pmc = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[2])):
    pmc[short(r['Kernel'])][r['Counter']] = float(r['MeanValue'])
print('| kernel | launches | µs | FETCH×2 + WRITE (MB) | MFMA busy cyc / SIMD (share of the launch at 2.4 GHz) | LDS conflict / active | waves parked |')
print('|---|---|---|---|---|---|---|')
for k, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    c = pmc.get(k, {})
    if not c:
        continue
    rd, wr = 2 * c.get('FETCH_SIZE', 0) * 1024 / 1e6, c.get('WRITE_SIZE', 0) * 1024 / 1e6
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
    lds = (100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']) if c.get('SQ_LDS_IDX_ACTIVE') else 0.0
    wait = (100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']) if c.get('SQ_WAVE_CYCLES') else 0.0
    print(f'| `{k}` | {calls} | {us:.1f} | {rd:.1f} + {wr:.1f} = {rd + wr:.1f} | {mf:,.0f} ({100 * mf / (us * 2400):.0f} %) | '
          f'{lds:.1f} % | {wait:.0f} % |')
