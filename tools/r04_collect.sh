#!/bin/bash
# Round-4 artefact collection on the GPU box (one gpurun call): smoke, bench line, rocprofv3 kernel stats + per-launch list of the same
# command, memory-side PMC passes (separate passes per counter; no trace domains beside --kernel-trace), phase breakdowns (plain /
# cooperative), exchange trace, full-length parity runs.  Output under gpurun_out/r04f/; the summaries are copied to profiles/.
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r04f; mkdir -p $O
( hostname; date +%T; MOSHII_COOP=1 python tools/chain_time.py 400 | tail -1; python tools/chain_time.py 400 | tail -1 ) > $O/box_probe.txt 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.txt
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 timeout 200 python tools/prof_chain.py 400 smplh > $O/phase_breakdown_one_workgroup.txt 2>&1
MOSHII_LIB=moshpp_amd/libmoshii_prof.so timeout 200 python tools/prof_chain.py 400 smplh > $O/phase_breakdown_cooperative.txt 2>&1
MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=6 timeout 200 python tools/coop_trace.py 200 > $O/coop_exchange_trace.txt 2>&1
timeout 300 python tools/coop_time.py 400 --groups=2,3,4,5,6,7,8 --fracs=0 > $O/coop_groups.txt 2>&1
( echo "# MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 python tools/prof_config3.py 40 80"; MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 timeout 200 python tools/prof_config3.py 40 80;
  echo; echo "# MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 python tools/prof_config3.py 40 80  (rank 0 of the group)"; MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 timeout 200 python tools/prof_config3.py 40 80;
  echo; echo "# MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 python tools/coop_trace.py 60 config3"; MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 timeout 200 python tools/coop_trace.py 60 config3 ) > $O/config3_phases.txt 2>&1
timeout 120 tools/bin/ubench_scope > $O/scope_ubench.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o bench -- python /root/repo/bench.py --no-cpu --no-stagei --no-config3 > /root/repo/$O/bench_line_under_rocprof.json 2> /root/repo/$O/rocprof_err.txt)
cp $O/stats/bench_kernel_stats.csv $O/ 2>/dev/null
python - <<'PY' > $O/bench_launches.txt 2>&1
import csv, glob
rows = []
for fn in glob.glob('gpurun_out/r04f/stats/*kernel_trace.csv'):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('# k_chain_solve / k_lbs launches of `python bench.py --no-cpu --no-stagei --no-config3` in dispatch order: duration (ms), grid, scratch B/lane, VGPR, AGPR, LDS')
for r in rows:
    n = r['Kernel_Name']
    if 'k_chain_solve' in n or 'k_lbs' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        print(f"{d:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7}  scratch {r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}  vgpr {r.get('VGPR_Count','?')} agpr {r.get('Accum_VGPR_Count','?')} lds {r.get('LDS_Block_Size', r.get('Group_Segment_Size','?'))}  {n[:70]}")
PY
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_$c -- python /root/repo/bench.py --no-cpu --no-stagei --no-strong --no-config3 --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> /root/repo/$O/pmc_${c}_err.txt)
  python - $O/pmc_$c > $O/pmc_$c.txt 2>&1 <<'PY'
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
print('# per k_chain_solve launch of `bench.py --steps 1 --warmup 1 --seeds 1000` (+ its sequential legs) in dispatch order: counter value as reported (KB), grid')
for r in rows:
    if 'k_chain_solve' in r['Kernel_Name']:
        print(f"{r['Counter_Name']:12s} {float(r['Counter_Value']):14.1f}  grid {r.get('Grid_Size', '?'):>7}  {r['Kernel_Name'][:70]}")
PY
  rm -rf $O/pmc_$c
done
timeout 600 python tools/full_parity_r04.py > $O/full_parity.txt 2>&1
timeout 600 python tools/config3_full_parity.py gpu > $O/config3_full_parity.txt 2>&1
( date +%T; MOSHII_COOP=1 python tools/chain_time.py 400 | tail -1 ) >> $O/box_probe.txt 2>&1
ls $O
