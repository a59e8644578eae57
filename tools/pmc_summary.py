"""Mean PMC counter values per kernel from rocprofv3 counter_collection csv files:
python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(list)
for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    with open(fn) as fh:
        for row in csv.DictReader(fh):
            if sub in row['Kernel_Name']:
                acc[(row['Kernel_Name'][:60], row['Counter_Name'])].append(float(row['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(f'{k:60s} {c:24s} n={len(v)} mean {sum(v) / len(v):.6g} sum {sum(v):.6g}')
