#!/bin/bash
# Round-6 artefact collection on the GPU box (one gpurun call): rocprofv3 kernel stats + per-launch list of the bench command,
# memory-side PMC passes (separate passes per counter; no trace domains beside --kernel-trace) over a chain step and over the LBS export
# (both bodies, still hands and all joints moving) -> profiles/r06_pmc.json (carries the library's source hash: bench.py --pmc-file refuses
# numbers of another build), SQ-side counters of the export, LBS timings / stamps / per-workgroup times, the bench line last.
# Output under gpurun_out/r06/; copy the summaries to profiles/.
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r06; mkdir -p $O
( hostname; date +%T ) > $O/box_probe.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o bench -- python /root/repo/bench.py --no-cpu --no-stagei --no-config3 > /root/repo/$O/bench_line_under_rocprof.json 2> /root/repo/$O/rocprof_err.txt)
cp $O/stats/bench_kernel_stats.csv $O/ 2>/dev/null
python - <<'PY' > $O/bench_launches.txt 2>&1
import csv, glob
rows = []
for fn in glob.glob('gpurun_out/r06/stats/*kernel_trace.csv'):
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
for hands in still moving; do for body in mesh shuffled; do LBS_HANDS=$hands LBS_BODY=$body LBS_CHECK=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh; done; done
echo "# the round-5 export kernel on the same box (moshpp_amd/libmoshii_r05lbs.so: HEAD~ of round 6's lbs_forward.hip), if present"
if [ -f moshpp_amd/libmoshii_r05lbs.so ]; then for hands in still moving; do MOSHII_LIB=$PWD/moshpp_amd/libmoshii_r05lbs.so LBS_HANDS=$hands LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 smplh; done; fi
LBS_HANDS=still LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 smplx
LBS_HANDS=still LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 50000 5 smplh
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 mano
echo "# MOSHII_LBS_STOP=8 (no still-joint shortcut) on the still-hands input"; LBS_HANDS=still LBS_BODY=mesh MOSHII_LBS_STOP=8 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# MOSHII_LBS_STOP=1 (prep + still + k-loop only)"; LBS_HANDS=still LBS_BODY=mesh MOSHII_LBS_STOP=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# MOSHII_LBS_STOP=2 (everything but the row stores)"; LBS_HANDS=still LBS_BODY=mesh MOSHII_LBS_STOP=2 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# MOSHII_LBS_STOP=16 (clock stamps of workgroup 0)"; LBS_HANDS=still LBS_BODY=mesh MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh
echo "# MOSHII_LBS_STOP=32 (start / end of every workgroup)"; LBS_HANDS=still LBS_BODY=mesh MOSHII_LBS_STOP=32 timeout 300 python tools/lbs_bench.py 4000 20 smplh
} 2>&1 | grep -v amdgpu.ids | cut -c1-900 > $O/lbs_timings.txt
for hands in still moving; do
(cd /tmp && LBS_HANDS=$hands LBS_BODY=mesh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/lbs_stats -o lbs -- python /root/repo/tools/lbs_bench.py 4000 20 smplh > /dev/null 2> /root/repo/$O/lbs_rocprof_err.txt)
cp $O/lbs_stats/lbs_kernel_stats.csv $O/lbs_kernel_stats_$hands.csv 2>/dev/null; rm -rf $O/lbs_stats
done
tools/bin/ubench_valu > $O/ubench_valu.txt 2>&1
# ---- SQ-side counters of the export (still hands, mesh body), three separate passes
{
echo "# k_lbs_export / k_lbs_prep / k_lbs_still, SMPL-H mesh-ordered body, still hands, F = 4000: rocprofv3 --kernel-trace --pmc <set> (three separate passes), per-launch means (tools/pmc_summary.py)"
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && LBS_HANDS=still LBS_BODY=mesh timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/sq$i -- python /root/repo/tools/lbs_bench.py 4000 3 smplh > /root/repo/$O/sq$i.log 2>&1)
  python tools/pmc_summary.py $O/sq$i k_lbs 2>&1
  rm -rf $O/sq$i $O/sq$i.log
done
} > $O/lbs_sq_counters.txt
# ---- memory-side PMC passes
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_chain_$c -- python /root/repo/bench.py --no-cpu --no-stagei --no-strong --no-config3 --no-sequential --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> /root/repo/$O/pmc_chain_${c}_err.txt)
  for body in mesh shuffled; do for hands in still moving; do
    (cd /tmp && LBS_HANDS=$hands LBS_BODY=$body timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/$O/pmc_lbs_${body}_${hands}_$c -- python /root/repo/tools/lbs_bench.py 4000 4 smplh > /dev/null 2> /root/repo/$O/pmc_lbs_${c}_err.txt)
  done; done
done
python - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, json, time, sys
sys.path.insert(0, '.')
from moshpp_amd import capi
O = 'gpurun_out/r06'
def rows(d):
    r = []
    for fn in glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True):
        r += list(csv.DictReader(open(fn)))
    r.sort(key=lambda x: int(x.get('Dispatch_Id', 0)))
    return r
out = {'source_hash': capi.load().moshii_source_hash().decode(), 'collected': time.strftime('%Y-%m-%d %H:%M'),
       'units': 'bytes; FETCH_SIZE x2 (gfx950 reports half the bytes of 16-byte-per-lane streams), counter values are KB'}
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
    i0 = idx[-1]
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
for body, btag in (('mesh', 'mesh_order'), ('shuffled', 'shuffled_ids')):
    for hands, htag in (('still', ''), ('moving', '_moving')):
        v = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            rs = rows(f'pmc_lbs_{body}_{hands}_{c}')
            for k in ('k_lbs_export', 'k_lbs_prep', 'k_lbs_still'):
                vals = [float(r['Counter_Value']) * 1024.0 for r in rs if k in r['Kernel_Name']]
                if vals:
                    v[(k, c)] = sum(vals[-4:]) / len(vals[-4:])      # the timed calls (the warm-up calls allocate and zero the per-call scratch)
                    print(f'# lbs {body} {hands} {k} {c}: {len(vals)} launches, mean of the last four {v[(k, c)] / 1e6:.1f} MB as reported')
        if len(v) == 6:
            tot = sum(2 * v[(k, 'FETCH_SIZE')] + v[(k, 'WRITE_SIZE')] for k in ('k_lbs_export', 'k_lbs_prep', 'k_lbs_still'))
            lbs[btag + htag] = {'bytes_per_call_at_4000_frames': tot, 'export_fetch_size_reported': v[('k_lbs_export', 'FETCH_SIZE')], 'export_write_size': v[('k_lbs_export', 'WRITE_SIZE')],
                                'prep_fetch_size_reported': v[('k_lbs_prep', 'FETCH_SIZE')], 'prep_write_size': v[('k_lbs_prep', 'WRITE_SIZE')],
                                'still_fetch_size_reported': v[('k_lbs_still', 'FETCH_SIZE')], 'still_write_size': v[('k_lbs_still', 'WRITE_SIZE')],
                                'ratio_to_algorithmic_372_7_MB': tot / 372.7e6}
if lbs:
    out['lbs'] = lbs
json.dump(out, open(f'{O}/r06_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_chain_* $O/pmc_lbs_*
# ---- Stage-I: per-kernel split of three solves of the bench problem
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/s1prof -o s1 -- python /root/repo/tools/stagei_time.py 3 > /root/repo/$O/stagei_time.txt 2>&1)
{ echo "# rocprofv3 --kernel-trace -- python tools/stagei_time.py 3 (three Stage-I solves of the bench problem: 12 frames, 53 markers, 10 betas, 925 unknowns, 36 dogleg iterations each), per kernel and launch shape (tools/stagei_trace_summary.py)"; grep '^rep' $O/stagei_time.txt | cut -c1-48; python tools/stagei_trace_summary.py $O/s1prof/s1_kernel_trace.csv 3 60; } > $O/stagei_kernel_summary.txt 2>&1
rm -rf $O/s1prof $O/stagei_time.txt
# ---- the bench line LAST, with the counters just collected on this very build in place
cp $O/r06_pmc.json profiles/r06_pmc.json
timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
ls $O
