import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows = [r for r in rows if 'k_chain_solve' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    print(f"grid {r.get('Grid_Size', r.get('Grid_Size_X')):>7}  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f} ms")
