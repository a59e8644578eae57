"""Per kernel and launch shape: calls, mean duration, ms per solve, share -- from a rocprofv3 --kernel-trace csv of tools/stagei_time.py.
usage: python tools/stagei_trace_summary.py <kernel_trace.csv> [solves=3] [rows=22]"""
import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
    agg[(name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in agg.values())
n=int(sys.argv[2]) if len(sys.argv)>2 else 3
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    print(f'{k[0]:22s} grid {k[1]:>7}x{k[2]:>3} wg {k[3]:>4}  n {len(v):4d}  avg {sum(v)/len(v)/1e3:7.1f} us  total {sum(v)/n/1e6:6.2f} ms/solve  {100*sum(v)/tot:4.1f}%')
print('total', tot/n/1e6, 'ms/solve')
