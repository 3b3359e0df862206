"""Per-launch timeline of one Newton-Schulz forward + backward from a rocprofv3 kernel trace of tools/ns_profile.py:
    python tools/ns_timeline.py <dir with *_kernel_trace.csv> > profiles/r3_ns_launch_timeline.csv
Takes the LAST repetition in the trace (clocks settled): start relative to its first kernel, duration, queue."""
import csv, glob, os, sys

files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a repetition starts at the covariance forward kernel (Gram panel kernel in mode 1 with centring)
starts = [i for i, n in enumerate(names) if 'bcnn_gram_panel_kernel' in n]
lo = starts[-1]
t0 = int(rows[lo]['Start_Timestamp'])
queues = {}
print('start_us,duration_us,queue,workgroups,lds_bytes,kernel')
for r in rows[lo:]:
    if 'hk::' not in r['Kernel_Name']:          # the script's closing prints (torch reductions): not part of the head
        break
    q = queues.setdefault(r['Queue_Id'], len(queues))
    wg = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1)) * int(r.get('Grid_Size_Z', 1)) // max(
        1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1)) * int(r.get('Workgroup_Size_Z', 1)))
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    print('%.1f,%.1f,%d,%d,%s,"%s"' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                    q, wg, r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '')), name))
