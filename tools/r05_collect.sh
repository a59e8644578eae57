#!/bin/bash
# Round-5 artefact collection on the GPU box (one gpurun call): rocprofv3 kernel stats + per-launch list of the same command,
# memory-side PMC passes (separate passes per counter; no trace domains beside --kernel-trace) over a chain step and over the LBS export
# on both bodies -> profiles/r05_pmc.json (carries the library's source hash: bench.py --pmc-file refuses numbers of another build),
# LBS timings / stamps / per-workgroup times.  Output under gpurun_out/r05/; copy the summaries to profiles/.
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r05; mkdir -p $O
( hostname; date +%T ) > $O/box_probe.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o bench -- python /root/repo/bench.py --no-cpu --no-stagei --no-config3 > /root/repo/$O/bench_line_under_rocprof.json 2> /root/repo/$O/rocprof_err.txt)
cp $O/stats/bench_kernel_stats.csv $O/ 2>/dev/null
python - <<'PY' > $O/bench_launches.txt 2>&1
import csv, glob
rows = []
for fn in glob.glob('gpurun_out/r05/stats/*kernel_trace.csv'):
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
# ---- LBS export alone: timings, kernel stats
{
for body in mesh shuffled; do LBS_BODY=$body LBS_CHECK=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh; done
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 smplx
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 50000 5 smplh
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 mano
echo "# MOSHII_LBS_STOP=1 (prep + k-loop only)"; LBS_BODY=mesh MOSHII_LBS_STOP=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# MOSHII_LBS_STOP=2 (everything but the row stores)"; LBS_BODY=mesh MOSHII_LBS_STOP=2 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# MOSHII_LBS_STOP=16 (clock stamps of workgroup 0)"; LBS_BODY=mesh MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh
echo "# MOSHII_LBS_STOP=32 (start / end of every workgroup)"; LBS_BODY=mesh MOSHII_LBS_STOP=32 timeout 300 python tools/lbs_bench.py 4000 20 smplh
} 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/lbs_timings.txt
(cd /tmp && LBS_BODY=mesh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/lbs_stats -o lbs -- python /root/repo/tools/lbs_bench.py 4000 20 smplh > /dev/null 2> /root/repo/$O/lbs_rocprof_err.txt)
cp $O/lbs_stats/lbs_kernel_stats.csv $O/ 2>/dev/null; rm -rf $O/lbs_stats
tools/bin/ubench_valu > $O/ubench_valu.txt 2>&1
# ---- PMC passes
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_chain_$c -- python /root/repo/bench.py --no-cpu --no-stagei --no-strong --no-config3 --no-sequential --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> /root/repo/$O/pmc_chain_${c}_err.txt)
  for body in mesh shuffled; do
    (cd /tmp && LBS_BODY=$body timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_lbs_${body}_$c -- python /root/repo/tools/lbs_bench.py 4000 4 smplh > /dev/null 2> /root/repo/$O/pmc_lbs_${c}_err.txt)
  done
done
python - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, json, time, sys
sys.path.insert(0, '.')
from moshpp_amd import capi
O = 'gpurun_out/r05'
def rows(d):
    r = []
    for fn in glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True):
        r += list(csv.DictReader(open(fn)))
    r.sort(key=lambda x: int(x.get('Dispatch_Id', 0)))
    return r
out = {'source_hash': capi.load().moshii_source_hash().decode(), 'collected': time.strftime('%Y-%m-%d %H:%M'),
       'units': 'bytes; FETCH_SIZE x2 (gfx950 reports half the bytes of 16-byte-per-lane streams), counter values are KB'}
# chain: the pass-1 launch (largest grid) of the timed step and the cooperative launches that follow it
ch = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rs = [r for r in rows(f'pmc_chain_{c}') if 'k_chain_solve' in r['Kernel_Name']]
    print(f'# chain {c}: per k_chain_solve launch in dispatch order (KB), grid')
    for r in rs:
        print(f"{c:12s} {float(r['Counter_Value']):14.1f}  grid {r.get('Grid_Size', '?'):>7}  {r['Kernel_Name'][:72]}")
    if not rs:
        continue
    gmax = max(int(r['Grid_Size']) for r in rs)
    idx = [i for i, r in enumerate(rs) if int(r['Grid_Size']) == gmax]
    i0 = idx[-1]                                    # the timed step's pass-1 launch (the warm-up step's comes first)
    rep = []
    for r in rs[i0 + 1:]:
        if int(r['Grid_Size']) == gmax:
            break
        rep.append(float(r['Counter_Value']) * 1024.0)
    ch[c] = {'pass1': float(rs[i0]['Counter_Value']) * 1024.0, 'repair': sum(rep), 'pass1_grid': gmax}
if 'FETCH_SIZE' in ch and 'WRITE_SIZE' in ch:
    n_chunks = ch['FETCH_SIZE']['pass1_grid'] // 256
    frames_pass1 = 4000 + n_chunks * 32
    out['chain'] = {'pass1_bytes_per_solved_frame': (2 * ch['FETCH_SIZE']['pass1'] + ch['WRITE_SIZE']['pass1']) / frames_pass1,
                    'repair_bytes_per_step': 2 * ch['FETCH_SIZE']['repair'] + ch['WRITE_SIZE']['repair'],
                    'pass1_fetch_size_reported': ch['FETCH_SIZE']['pass1'], 'pass1_write_size': ch['WRITE_SIZE']['pass1'],
                    'frames_solved_in_pass1': frames_pass1, 'command': 'bench.py --no-cpu --no-stagei --no-strong --no-config3 --no-sequential --steps 1 --warmup 1 --seeds 1000'}
lbs = {}
for body, tag in (('mesh', 'mesh_order'), ('shuffled', 'shuffled_ids')):
    v = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        rs = rows(f'pmc_lbs_{body}_{c}')
        for k in ('k_lbs_export', 'k_lbs_prep'):
            vals = [float(r['Counter_Value']) * 1024.0 for r in rs if k in r['Kernel_Name']]
            if vals:
                v[(k, c)] = sum(vals[-4:]) / len(vals[-4:])      # the timed calls (the warm-up calls allocate and zero the per-call scratch)
                print(f'# lbs {body} {k} {c}: {len(vals)} launches, mean of the last four {v[(k, c)] / 1e6:.1f} MB as reported')
    if len(v) == 4:
        tot = 2 * v[('k_lbs_export', 'FETCH_SIZE')] + v[('k_lbs_export', 'WRITE_SIZE')] + 2 * v[('k_lbs_prep', 'FETCH_SIZE')] + v[('k_lbs_prep', 'WRITE_SIZE')]
        lbs[tag] = {'bytes_per_call_at_4000_frames': tot, 'export_fetch_size_reported': v[('k_lbs_export', 'FETCH_SIZE')], 'export_write_size': v[('k_lbs_export', 'WRITE_SIZE')],
                    'prep_fetch_size_reported': v[('k_lbs_prep', 'FETCH_SIZE')], 'prep_write_size': v[('k_lbs_prep', 'WRITE_SIZE')],
                    'ratio_to_algorithmic_372_7_MB': tot / 372.7e6}
if lbs:
    out['lbs'] = lbs
json.dump(out, open(f'{O}/r05_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_chain_* $O/pmc_lbs_*
# ---- the bench line LAST, with the counters just collected on this very build in place (bench.py --pmc-file defaults to profiles/r05_pmc.json
#      and refuses a file of another source hash)
cp $O/r05_pmc.json profiles/r05_pmc.json
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
# ---- where the default chain mode switches; the frame's phases (profile build: python -m moshpp_amd.build --profile); the clock under load
timeout 900 python tools/auto_threshold.py 2>&1 | grep -v amdgpu.ids > $O/auto_threshold.txt
if [ -f moshpp_amd/libmoshii_prof.so ]; then
  MOSHII_LIB=$PWD/moshpp_amd/libmoshii_prof.so timeout 300 python tools/prof_chain.py 400 smplh 2>&1 | grep -v amdgpu.ids > $O/phase_breakdown_cooperative.txt
  MOSHII_COOP=1 MOSHII_LIB=$PWD/moshpp_amd/libmoshii_prof.so timeout 300 python tools/prof_chain.py 400 smplh 2>&1 | grep -v amdgpu.ids > $O/phase_breakdown_one_workgroup.txt
fi
ls $O
