#!/bin/bash
# Round-3 artefact collection on the GPU box (one gpurun call): smoke, bench line, rocprofv3 kernel stats + trace of the same command,
# memory-side PMC passes over a bench step, in-kernel phase breakdown.  Output under gpurun_out/r03f/; summaries are copied to profiles/.
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r03f; mkdir -p $O
# box probe: the pool's boxes are not all alike (two collections of this round ran ~15 % slower on every kernel than the runs around
# them with the same binary) -- record a 400-frame chain (266 us/frame on the usual box) before and after the collection
( hostname; date +%T; python tools/chain_time.py 400 | tail -1 ) > $O/box_probe.txt 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.txt
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python -m moshpp_amd.build --profile > /dev/null 2>&1
MOSHII_LIB=moshpp_amd/libmoshii_prof.so timeout 200 python tools/prof_chain.py 400 smplh > $O/phase_breakdown.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o bench -- python /root/repo/bench.py --no-cpu --no-stagei --no-config3 > /root/repo/$O/bench_line_under_rocprof.json 2> /root/repo/$O/rocprof_err.txt)
cp $O/stats/bench_kernel_stats.csv $O/ 2>/dev/null
python - <<'PY' > $O/bench_launches.txt 2>&1
import csv, glob
rows = []
for fn in glob.glob('gpurun_out/r03f/stats/*kernel_trace.csv'):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('# k_chain_solve / k_verify launches of `python bench.py --no-cpu --no-stagei --no-config3` in dispatch order: duration (ms), grid, scratch B/lane, VGPR, AGPR, LDS')
for r in rows:
    n = r['Kernel_Name']
    if 'k_chain_solve' in n or 'k_lbs' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        print(f"{d:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7}  scratch {r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}  vgpr {r.get('VGPR_Count','?')} agpr {r.get('Accum_VGPR_Count','?')} lds {r.get('LDS_Block_Size', r.get('Group_Segment_Size','?'))}  {n[:60]}")
PY
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_$c -- python /root/repo/bench.py --no-cpu --no-stagei --no-strong --no-sequential --no-config3 --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> /root/repo/$O/pmc_${c}_err.txt)
  python tools/pmc_summary.py $O/pmc_$c k_chain_solve > $O/pmc_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
( date +%T; python tools/chain_time.py 400 | tail -1 ) >> $O/box_probe.txt 2>&1
ls $O
