import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows = [r for r in rows if 'k_lbs' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-30:]
prev_end = None
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].split('(')[0][-16:]
    print(f"{nm:18s} dur {(en-st)/1e3:8.1f} us   gap before {((st-prev_end)/1e3 if prev_end else 0):6.1f} us")
    prev_end = en
