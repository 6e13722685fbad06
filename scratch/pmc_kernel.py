"""Mean PMC counter values per kernel for the last launches: python scratch/pmc_kernel.py <dir> <kernel> [last_n]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
kern = sys.argv[2]; last = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if kern in r['Kernel_Name']:
        rows[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for c, v in sorted(rows.items()):
    v = [x for _, x in sorted(v)][-last:]
    print('%-28s %14.1f' % (c, sum(v) / len(v)))
