"""Aggregate rocprofv3 counter-collection CSVs (one directory per --pmc pass) into Kernel,Counter,Dispatches,MeanValue
rows: per dispatch the values of a counter are summed over its instances / dimensions, then averaged over dispatches.
    python tools/pmc_summary.py DIR [DIR ...] [--only hk::] > profiles/rN_pool_kernels_pmc.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else ''
if '--only' in sys.argv:
    dirs.remove(only)
acc = defaultdict(lambda: defaultdict(float))          # (kernel, counter) -> dispatch -> sum
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', '')).strip()
            if only and only not in k:
                continue
            acc[(k, r['Counter_Name'])][(f, r['Dispatch_Id'])] += float(r['Counter_Value'])
w = csv.writer(sys.stdout)
w.writerow(['Kernel', 'Counter', 'Dispatches', 'MeanValue'])
for (k, c), v in sorted(acc.items()):
    w.writerow([k, c, len(v), sum(v.values()) / len(v)])
