"""Kernel timeline of the last alignment in a rocprofv3 --kernel-trace CSV: python scratch/trace_passes.py <dir> [max rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('symmicp::', '').replace('void ', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith('k_search_packet') or r[2].startswith('k_search_walk<false>') and False]
i0 = idx[-1] if idx else 0
t_prev = rows[i0][0]
for s, e, n in rows[i0:i0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 70)]:
    print("%-40s %8.1f us  gap %6.1f" % (n[:40], (e - s) / 1e3, (s - t_prev) / 1e3)); t_prev = e
