"""Per-pass kernel durations (us) from a rocprofv3 --kernel-trace CSV: python scratch/per_pass.py <dir> [first [count]]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 12
names = ('k_search_cells', 'k_search_walk', 'k_accumulate', 'k_final_reduce')
seq = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('symmicp::', '').replace('void ', '').split('<')[0]
    if n in names:
        seq[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for n in seq: seq[n].sort()
N = len(seq['k_accumulate'])
print('pass  cells   walk    acc  final   span  host-gap')
for i in range(first, min(N, first + count)):
    k = [seq[n][i] for n in names]
    gap = (seq['k_search_cells'][i + 1][0] - k[3][1]) / 1e3 if i + 1 < N else 0
    print('%4d %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f' % ((i,) + tuple((x[1] - x[0]) / 1e3 for x in k) + ((k[3][1] - k[0][0]) / 1e3, gap)))
