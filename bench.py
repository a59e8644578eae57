#!/usr/bin/env python
"""Stage-II throughput bench (BASELINE.json metric: solved mocap frames/sec).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full Stage-II pass over one synthetic sequence of BASELINE config[1] (4000-frame SMPL-H, 53 body markers,
fixed betas) with observations and outputs resident in HBM: `moshii_sequence_solve` cuts the sequence into one chunk per CU,
solves the chunks concurrently (each starts --chunk-warmup frames early), verifies every hand-off against its predecessor's
end state on the device and re-solves the chunks that miss --verify-tol exactly -- all of that is inside the timed region.
How long the repairs take depends on the motion, so the timed steps CYCLE through --seeds (six seeded sequences of the same
shape; step k solves sequence k mod 6) and `value` is ALL timed frames / ALL timed seconds (since round 4; `seeds` holds every
seed's rate, min and max beside it; `median_over_seeds` is what rounds 1-3 printed as `value`).
`--mode sequential` times the reference's literal frame order instead (one chain = one workgroup).

N > 1 (torch.distributed.run, one rank per GPU): every rank runs the same steps on its own copies -- the path has no
data-path collective -- and the job-level frames/s is reported: "scaling": "weak" (replicas).  The partition north_star
asks for is measured beside it at every N, on FIXED jobs (object `strong`): 32 sequences dealt to the ranks by
parallel.partition_units (BASELINE config 3's shape; N = 1 is the `many_sequences` figure), and one 50 000-frame sequence cut
into frame ranges with verified hand-offs (parallel.solve_sequence_sharded; config 5's shape).  `--scaling strong` makes the
32-sequence job the headline (`value`, `scaling: "strong"`).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline           dominant kernel k_chain_solve against the f64 vector peak (it is latency/ALU-bound, not HBM-bound)
  roofline_lbs       the full-mesh LBS export kernel against the HBM peak
  cpu_baseline       the NumPy oracle ("port") on bounded samples, host cores of this box: one core in its lean mode (`value`),
                     one core doing the reference's own amount of work per iteration (`reference_cost`: full-mesh forward + dense
                     3V x 3K Jacobian, smpl_fast_derivatives.py:250-256), and one sequence per core on `all_cores`
  parity             GPU result of the timed mode vs the oracle's sequential chain on that sample
  sequential_chain   the literal frame order on the GPU (one workgroup) and the timed mode's deviation from it over ALL frames
  incl_host_staging  the same solve through host buffers (observations in, results out over PCIe)
  many_sequences     32 sequences in one call: the dominant kernel with every CU busy
  config3            BASELINE config 3 as stated (32 x 4000 SMPL-X frames, 89 markers, fingers + jaw + 80 expressions free: 194 unknowns)
  strong             the fixed jobs above, at this N
  stagei             Stage-I on 12 frames / 53 markers / 10 betas: GPU seconds (dense and arrow-structured solver), iterations, and
                     the NumPy oracle's seconds + differences on the same problem
The reference / CPU / Stage-I legs run at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (AMD spec sheet; half the 157.3 TF FP32 rate of MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


LINE_LIMIT = 3600   # bytes (the driver keeps the last 4-8 KB of stdout; round 6's first line came to 3968): the driver keeps the tail of stdout; the round-5 line (20 KB) was cut and could not be parsed


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """The ONE stdout line: the contract's keys + the few numbers each extra leg is quoted by (<= LINE_LIMIT bytes, strict JSON).
    Everything else -- per-seed tables, notes, workloads spelled out -- goes to bench_detail.json / stderr (`write_detail`)."""
    out = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                       'vs_baseline', 'dtype', 'data'))
    cfg = full.get('config', {})
    out['config'] = dict(_pick(cfg, ('mode', 'frames_per_gpu', 'markers', 'free_vars_step1', 'free_vars_step2', 'sequences_per_gpu', 'parallelism')),
                         workload=str(cfg.get('workload', ''))[:140])
    out['config'] = {'workload': out['config'].pop('workload'), **out['config']}
    if 'rccl' in full:
        out['rccl'] = full['rccl']
    rf = full.get('roofline')
    if isinstance(rf, dict):
        out['roofline'] = _pick(rf, ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'step_ms_hip_events',
                                     'algorithmic_gflop_per_step'))
        out['roofline']['traffic_is'] = str(rf.get('traffic_is', ''))[:48]
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        out['cpu_baseline'] = dict(_pick(cb, ('value', 'unit', 'cores', 'kind')), sample=str(cb.get('sample', ''))[:80])
        if isinstance(cb.get('reference_cost'), dict):
            out['cpu_baseline']['reference_cost'] = cb['reference_cost'].get('value')
        if isinstance(cb.get('all_cores'), dict) and 'value' in cb['all_cores']:
            out['cpu_baseline']['all_cores'] = _pick(cb['all_cores'], ('value', 'cores'))
    rl = full.get('roofline_lbs')
    if isinstance(rl, dict):
        out['roofline_lbs'] = _pick(rl, ('kernel', 'bound', 'body', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_ms',
                                         'algorithmic_bytes', 'error'))
        if 'traffic_is' in rl:
            out['roofline_lbs']['traffic_is'] = str(rl['traffic_is'])[:48]
        for leg in ('all_joints_moving', 'shuffled_vertex_ids', 'shuffled_vertex_ids_all_joints_moving'):
            if isinstance(rl.get(leg), dict):
                out['roofline_lbs'][leg] = _pick(rl[leg], ('frac', 'kernel_ms', 'traffic'))
    pe = full.get('parity_every_frame')
    if isinstance(pe, dict):
        out['parity'] = _pick(pe, ('frames_checked', 'frames_outside_tolerance', 'frames_parted_on_a_knife_edge', 'frames_over_1e-4_rad',
                                   'max_dev_on_well_conditioned_frames_rad', 'worst_sequence_marker_rmse_vs_oracle_m', 'all_ok'))
        if isinstance(pe.get('configs'), dict):
            out['parity']['configs'] = {k.replace('_4000_frames', '').replace('_frames', ''): (_pick(v, ('frames', 'frames_outside_tolerance', 'frames_over_1e-4_rad', 'marker_rmse_vs_oracle_m', 'ok')) if isinstance(v, dict) else str(v)[:120])
                                        for k, v in pe['configs'].items()}
    pl = full.get('parity')
    if isinstance(pl, dict):
        out['parity_live_oracle'] = _pick(pl, ('frames', 'max_abs_pose_diff_rad', 'marker_rmse_m', 'frames_outside_tolerance_vs_oracle_all_seeds'))
    sq = full.get('sequential_chain')
    if isinstance(sq, dict):
        out['sequential_chain'] = _pick(sq, ('frames_per_s', 'us_per_frame', 'kernel'))
        if isinstance(sq.get('one_workgroup'), dict):
            out['sequential_chain']['one_workgroup_us_per_frame'] = sq['one_workgroup'].get('us_per_frame')
    c3 = full.get('config3')
    if isinstance(c3, dict):
        out['config3'] = _pick(c3, ('frames_per_s', 'frames', 'kernel', 'marker_rmse_m', 'error'))
        if isinstance(c3.get('roofline'), dict):
            out['config3']['frac'] = c3['roofline'].get('frac')
    ms = full.get('many_sequences')
    if isinstance(ms, dict):
        out['many_sequences'] = _pick(ms, ('sequences', 'frames_per_s', 'ms'))
        if isinstance(ms.get('roofline'), dict):
            out['many_sequences']['frac'] = ms['roofline'].get('frac')
    st = full.get('strong')
    if isinstance(st, dict):
        out['strong'] = {k: _pick(v, ('frames', 'frames_per_s', 'ms', 'error')) for k, v in st.items() if isinstance(v, dict)}
    for k in ('one_gpu_same_job', 'speedup_vs_one_gpu_same_job', 'seed_min', 'seed_max', 'median_over_seeds', 'default_mode',
              'speedup_vs_cpu_port', 'speedup_vs_cpu_reference_cost', 'speedup_vs_sequential_chain'):
        if k in full:
            out[k] = _pick(full[k], ('frames_per_s', 'ms', 'speedup')) if isinstance(full[k], dict) else full[k]
    if isinstance(full.get('incl_host_staging'), dict):
        out['incl_host_staging_frames_per_s'] = full['incl_host_staging'].get('frames_per_s')
    s1 = full.get('stagei')
    if isinstance(s1, dict):
        out['stagei'] = _pick(s1, ('seconds', 'dogleg_iterations', 'launches', 'max_abs_betas_diff', 'error'))
    if isinstance(full.get('seeds'), dict):
        out['seeds_frames_per_s'] = {k: v.get('frames_per_s') for k, v in full['seeds'].items() if isinstance(v, dict)}
    out['detail'] = full.get('detail_file', 'bench_detail.json (+ stderr)')

    def short(x, key=None):   # six significant digits on the line (the detail file keeps every digit); the contract's own numbers untouched
        if isinstance(x, dict):
            return {k: short(v, k) for k, v in x.items()}
        if isinstance(x, list):
            return [short(v) for v in x]
        if isinstance(x, float) and key not in ('value', 'ms_per_step'):
            return float(f'{x:.6g}')
        return x
    out = short(out)
    line = json.dumps(out, allow_nan=False, default=float)
    # never over the limit: drop the optional blocks, least important first
    for k in ('seeds_frames_per_s', 'parity_live_oracle', 'incl_host_staging_frames_per_s', 'stagei', 'strong', 'sequential_chain',
              ('parity', 'configs'), 'many_sequences', 'config3'):
        if len(line) < LINE_LIMIT:
            break
        if isinstance(k, tuple):
            if isinstance(out.get(k[0]), dict):
                out[k[0]].pop(k[1], None)
        else:
            out.pop(k, None)
        line = json.dumps(out, allow_nan=False, default=float)
    assert len(line) < LINE_LIMIT, len(line)
    return line


def _finite(o):
    """NaN / inf are not JSON: null them (a leg that produced one says so in the detail file's `non_finite` list)."""
    bad = []

    def walk(x, path):
        if isinstance(x, dict):
            return {k: walk(v, path + '.' + str(k)) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [walk(v, path + '[]') for v in x]
        if isinstance(x, (float, np.floating)):
            if not np.isfinite(x):
                bad.append(path)
                return None
            return float(x)
        if isinstance(x, np.integer):
            return int(x)
        if isinstance(x, np.bool_):
            return bool(x)
        return x
    r = walk(o, '')
    if bad:
        r['non_finite'] = bad
    return r


def write_detail(full):
    """The whole result: bench_detail.json beside bench.py (and under gpurun_out/ when that exists: it is what travels back from a
    GPU box), one line on stderr.  Returns the file name written (or None)."""
    txt = json.dumps(full)
    print('bench detail: ' + txt, file=sys.stderr, flush=True)
    wrote = None
    for d in (os.path.join(ROOT, 'gpurun_out'), ROOT):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, 'bench_detail.json'), 'w') as fh:
                    fh.write(txt + '\n')
                wrote = wrote or os.path.relpath(os.path.join(d, 'bench_detail.json'), ROOT)
        except OSError:
            pass
    return wrote


def solver_flops(K, Nv, n1, n2, NP, npose, nobs_mean, iters, fevals):
    """Algorithmic FLOPs of the solve (SURVEY.md 8(d) / DESIGN.md 5): forward 2*3Nv*9(K-1) + 2*Nv*K*12 per residual
    evaluation; per Jacobian 3Nv*n*30 + normal equations 2*m*n^2/2 + factorisation n^3/3, m = 3*nobs + (npose+1) + NP."""
    fwd = 2.0 * 3 * Nv * 9 * (K - 1) + 2.0 * Nv * K * 12
    m = 3.0 * nobs_mean + (npose + 1 if npose else 0) + NP
    n = 3 + 0.5 * (n1 + n2)
    per_iter = 3.0 * Nv * n * 30 + m * n * n + n ** 3 / 3.0
    return float(fevals) * fwd + float(iters) * per_iter


def _cpu_chain_worker(a):
    """One oracle chain on `frames` frames of the seeded workload (a process of the all-cores CPU leg)."""
    seed, frames, markers = a
    sys.path.insert(0, ROOT)
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=1)
    except Exception:
        pass
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    job = workload.make_job('smplh', n_frames=frames, n_markers=markers, seed=seed)
    m, pr, closest, coef = oracle_setup(job)
    t0 = time.perf_counter()
    ref = so.stageii_chain(m, pr, closest, coef, job['obs'], job['vis'], 'smplh')
    return len(ref['frame_ids']), time.perf_counter() - t0


def _cpu_parity_worker(a):
    """The oracle's sequential chain on the first `sample` frames of the FULL seeded workload (the same job the GPU timed):
    fullpose and simulated markers per solved frame -- the checker of the `parity` block, one process per timed seed."""
    seed, frames, markers, sample = a
    sys.path.insert(0, ROOT)
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=1)
    except Exception:
        pass
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    job = workload.make_job('smplh', n_frames=frames, n_markers=markers, seed=seed)
    m, pr, closest, coef = oracle_setup(job)
    ref = so.stageii_chain(m, pr, closest, coef, job['obs'][:sample], job['vis'][:sample], 'smplh')
    return seed, np.asarray(ref['frame_ids']), np.asarray(ref['fullpose']), [np.asarray(x) for x in ref['markers_sim']]


def oracle_setup(job):
    from oracle import stageii_oracle as so
    sm = job['sm']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    pr = so.prepare_gmm_prior(job['seq']['gmm'], 63)
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'])
    return m, pr, closest, coef


def strong_job_shares(n_sequences, frames, world):
    """The fixed many-sequence job: which sequences (indices into the seed cycle) each rank solves (LPT by frame count)."""
    from moshpp_amd.parallel import partition_units
    return partition_units([float(frames)] * n_sequences, world)


def load_pmc(path, lib_hash):
    """PMC numbers collected by tools/r06_collect.sh, or (None, why): never numbers of another build."""
    try:
        with open(path) as fh:
            d = json.load(fh)
    except Exception as e:
        return None, f'no PMC file ({path}: {e!r})'
    if d.get('source_hash') != lib_hash:
        return None, f'PMC file {os.path.basename(path)} was collected on source hash {d.get("source_hash")}, the loaded library is {lib_hash}: not used'
    return d, f'rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes (x2 on FETCH_SIZE for 16-byte-per-lane streams, MI355X_MICROARCH.md), {os.path.relpath(path, ROOT)}, collected {d.get("collected", "?")} on source hash {lib_hash}'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=int(os.environ.get('WORLD_SIZE', '1')), help='ranks = GPUs (default: WORLD_SIZE under a launcher, else 1)')
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=4000)
    ap.add_argument('--markers', type=int, default=53)
    ap.add_argument('--mode', choices=('chunked', 'sequential'), default='chunked')
    ap.add_argument('--scaling', choices=('auto', 'weak', 'strong'), default='auto',
                    help='strong: the fixed many-sequence job (sequences dealt to the ranks) is the headline; weak: every rank its own copies of '
                         'the config[1] sequence (replicas); auto = config[1] on one GPU, strong on several')
    ap.add_argument('--seeds', default='1000,123,71,5,2024,7', help='the timed steps cycle through sequences generated from these seeds')
    ap.add_argument('--chunks', type=int, default=0, help='chunks per sequence (0 = one per CU)')
    ap.add_argument('--chunk-warmup', type=int, default=32)
    # hand-off tolerance: 1e-9 rad / m yields the same stitched result as 1e-11 (max deviation from the sequential chain 1.4e-9
    # rad on all 4000 frames, re-measured below every run) with fewer repairs of warm-ups that were converged to 1e-10
    ap.add_argument('--verify-tol', type=float, default=1e-9)
    ap.add_argument('--cpu-sample', type=int, default=240, help='frames of the workload timed on the CPU oracle (one core, lean mode)')
    ap.add_argument('--cpu-procs', type=int, default=0, help='processes of the all-cores CPU leg (0 = min(host cores, 32))')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-stagei', action='store_true', help='skip the Stage-I leg')
    ap.add_argument('--no-sequential', action='store_true', help='skip the one-workgroup sequential reference run')
    ap.add_argument('--no-strong', action='store_true', help='skip the fixed-job (strong scaling) legs')
    ap.add_argument('--strong-sequences', type=int, default=256, help='distinct captures in the fixed many-sequence job (256 x 4000 frames: ~2 s on one GPU)')
    ap.add_argument('--long-frames', type=int, default=50000)
    ap.add_argument('--no-config3', action='store_true', help='skip the BASELINE config 3 leg (32 x 4000 SMPL-X frames, 194 unknowns: ~30 s)')
    ap.add_argument('--config3-sequences', type=int, default=32)
    ap.add_argument('--config3-frames', type=int, default=4000)
    ap.add_argument('--lbs-frames', type=int, default=4000)   # the whole solved sequence: that is what a mesh export writes
    ap.add_argument('--pmc-file', default=os.path.join(ROOT, 'profiles', 'r06_pmc.json'),
                    help='HBM traffic from rocprofv3 PMC passes (tools/r06_collect.sh); used only if its source hash is the loaded library\'s')
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    # stdout carries exactly one line, the result: whatever native libraries print on the way (gloo announces its connections on
    # stdout when the hand-off group of the strong-scaling leg is created) goes to stderr
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(args.gpus, 1):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the Stage-II path has no CPU fallback')
    # development aid: MOSHII_BENCH_ONE_GPU=1 runs every rank on cuda:0 with gloo as the process group -- the N > 1 code paths
    # (partition of the fixed job, sharded long sequence, max-over-ranks timing) on a one-GPU box; the numbers mean nothing
    one_gpu = os.environ.get('MOSHII_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_probe = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            # prove the N ranks on real devices before anything is timed: the backend is RCCL, and a sum over the ranks of (1, own device
            # index + 1) taken ON the devices gives the rank count and tells distinct GPUs from N processes on one
            assert dist.get_backend() == 'nccl', dist.get_backend()
            probe = torch.tensor([1.0, float(local_rank + 1)], device=torch.device('cuda', local_rank))
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            rccl_ranks, dev_sum = int(probe[0].item()), int(probe[1].item())
            assert rccl_ranks == world, (rccl_ranks, world)
            rccl_probe = {'backend': 'nccl (RCCL)', 'ranks_seen_by_an_all_reduce': rccl_ranks,
                          'distinct_devices': bool(dev_sum == world * (world + 1) // 2 and torch.cuda.device_count() >= world)}

    from moshpp_amd import capi, workload
    lib = capi.load()
    capi.check(lib.moshii_set_device(local_rank))
    dev = torch.device('cuda', local_rank)
    seeds = [int(x) for x in args.seeds.split(',') if x.strip()]

    # ---- workload: BASELINE config[1], one resident sequence per seed (same model / betas layout, different motion)
    jobs = {sd: workload.make_job('smplh', n_frames=args.frames, n_markers=args.markers, seed=sd) for sd in seeds}
    solvers = {sd: workload.make_solver(jobs[sd]) for sd in seeds}
    seqs = {sd: workload.DeviceSequence(jobs[sd], solvers[sd], dev) for sd in seeds}
    job, solver, ds = jobs[seeds[0]], solvers[seeds[0]], seqs[seeds[0]]
    sm = job['sm']
    F, M = job['vis'].shape
    reports = {sd: [] for sd in seeds}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device='cpu' if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(x):
        if dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device='cpu' if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # ---- the fixed many-sequence job (strong scaling; also the headline with --scaling strong)
    def strong_many():
        """The fixed many-sequence job: --strong-sequences DISTINCT captures of the seed-1000 subject (sequences of one subject share the
        model / prior handles of a call; capture i is motion seed 5000 + i through workload.make_capture -- the job waits for its hardest
        sequence, as a real batch does), dealt to the ranks by longest-processing-time; a rank solves its share in ONE
        moshii_sequence_solve call.  On several GPUs rank 0 first solves the WHOLE job alone (the other ranks wait), so that the line
        carries the one-GPU time of the same job on the same box.  Returns (frames solved here, seconds here, sequences here,
        one-GPU seconds or None, one-GPU frames)."""
        shares = strong_job_shares(args.strong_sequences, F, world)
        share = shares[rank]
        stream = torch.cuda.current_stream().cuda_stream

        def solve(idx, reps=1, warm=1, sync=None):
            """`warm` untimed passes over the share (>= 1: allocations inside the library), then `reps` timed ones (a pass = one step of the
            N > 1 headline); sync(): the ranks' barrier in front of the timed passes.  Returns seconds PER PASS and frames per pass."""
            copies = [workload.DeviceSequence(workload.make_capture(job, solver, 5000 + i), solver, dev) for i in idx]
            per_seq_chunks = max(8, 256 // max(len(copies), 1))      # keep every CU carrying a chain when a rank has few sequences
            for _ in range(max(1, warm)):
                workload.solve_many_chunked(copies, stream, num_chunks=per_seq_chunks)
            torch.cuda.synchronize()
            if sync is not None:
                sync()
            t0 = time.perf_counter()
            for _ in range(max(1, reps)):
                workload.solve_many_chunked(copies, stream, num_chunks=per_seq_chunks)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / max(1, reps)
            return dt, sum(int((c.results()['status'] != 1).sum()) for c in copies)

        t_one, n_one = None, 0
        if world > 1:
            if rank == 0:
                t_one, n_one = solve(list(range(args.strong_sequences)))
            barrier()
        n_local = len(share)
        t_here, n_here = 0.0, 0
        # the N > 1 headline: --warmup untimed and --steps timed passes over the job, a barrier on both sides of the timed ones (a rank
        # without a share still meets the barriers); on one GPU the leg is an extra of the default line: one warm-up, one pass
        reps, warm = (args.steps, args.warmup) if headline_is_strong else (1, 1)
        if n_local:
            t_here, n_here = solve(share, reps=reps, warm=warm, sync=barrier)
        else:
            barrier()
        barrier()
        return n_here, t_here, n_local, t_one, n_one, reps, max(1, warm)

    def strong_long():
        """One long sequence over the ranks by frame ranges (host buffers: the boundary rows travel through the process group)."""
        from moshpp_amd.parallel import solve_sequence_sharded
        jl = workload.make_job('smplh', n_frames=args.long_frames, n_markers=args.markers, seed=seeds[0])
        sl = workload.make_solver(jl)
        calls = []

        def solve_range(a, b, init):
            calls.append((a, b))
            return sl.solve(jl['obs'][a:b], jl['vis'][a:b], chain_mode='chunked', verify_tol=args.verify_tol, init=init)

        gl = None
        if dist is not None:
            gl = dist.new_group(backend='gloo')   # object collectives of the hand-off check: host side
        solve_range(0, min(600, args.long_frames), None); calls.clear()      # untimed first call
        barrier()
        t0 = time.perf_counter()
        out, info = solve_sequence_sharded(solve_range, args.long_frames, dist=_GroupView(dist, gl) if dist is not None else None,
                                           warmup=args.chunk_warmup, verify_tol=args.verify_tol)
        torch.cuda.synchronize()
        t_here = time.perf_counter() - t0
        n_here = int((np.asarray(out['status']) != 1).sum())
        barrier()
        return n_here, t_here, info, len(calls)

    class _GroupView:
        """torch.distributed with the object collectives routed to a gloo group (parallel.solve_sequence_sharded's interface)."""
        def __init__(self, d, g): self.d, self.g = d, g
        def is_initialized(self): return True
        def get_world_size(self): return self.d.get_world_size()
        def get_rank(self): return self.d.get_rank()
        def all_gather_object(self, out, obj): return self.d.all_gather_object(out, obj, group=self.g)
        def gather_object(self, obj, out, dst=0): return self.d.gather_object(obj, out, dst=dst, group=self.g)

    def step(k):
        sd = seeds[k % len(seeds)]
        stream = torch.cuda.current_stream().cuda_stream
        if args.mode == 'chunked':
            reports[sd].append(seqs[sd].solve_chunked(stream, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol))
        else:
            seqs[sd].solve_sequential(stream)

    for k in range(max(args.warmup, 0)):
        step(k)
    for sd in seeds:
        reports[sd].clear()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(k)
        ev[k][1].record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    barrier()
    step_ms = np.array([a.elapsed_time(b) for a, b in ev])
    t_max = allmax(t_local)
    used = [sd for i, sd in enumerate(seeds) if i < args.steps]
    res = {sd: seqs[sd].results() for sd in used}
    solved_of = {sd: int((res[sd]['status'] != 1).sum()) for sd in used}
    frames_timed = sum(solved_of[seeds[k % len(seeds)]] for k in range(args.steps))
    total_frames = allsum(frames_timed)
    aggregate = total_frames / t_max
    ms_per_step = 1e3 * t_max / args.steps
    per_seed = {}
    for i, sd in enumerate(used):
        ms = step_ms[i::len(seeds)]
        rp = reports[sd][-1] if reports[sd] else None
        per_seed[str(sd)] = {'frames_per_s': round(world * solved_of[sd] / (float(ms.mean()) * 1e-3), 1), 'ms_per_step': round(float(ms.mean()), 3),
                             'steps': int(len(ms)), **({'n_repaired': rp['n_repaired'], 'repair_rounds': rp['repair_rounds']} if rp else {})}
    rates = np.array([v['frames_per_s'] for v in per_seed.values()])
    median_rate = float(np.median(rates))
    value = float(aggregate)     # the job rate: all timed frames / all timed seconds (max over ranks); the median over seeds rides beside it

    out = res[seeds[0]]
    status, iters = out['status'], out['iters']
    solved_mask = status != 1
    solved = int(solved_mask.sum())
    name, lds, thr = capi.last_launch_info()
    rep = reports[seeds[0]][-1] if reports[seeds[0]] else None
    how = (f'chunked: {rep["n_chunks"]} concurrent chunks (1 workgroup each), {rep["warmup"]}-frame warm-up overlap, hand-offs '
           f'verified to {rep["verify_tol"]:g} and repaired exactly by cooperative sweeps ({name})' if rep else f'exact sequential chain ({name})')
    result = {
        'metric': 'solved mocap frames/sec (Stage-II)', 'value': round(value, 2), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        **({'rccl': rccl_probe} if rccl_probe else {}),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'BASELINE config[1]: {F}-frame SMPL-H sequence, {M} body markers, fixed betas; {how}; the steps cycle '
                               f'through {len(used)} seeded sequences, value = all timed frames / all timed seconds'
                               + ('' if world == 1 else f'; {world} replicas (every rank its own copies: no data-path collective)'),
                   'mode': args.mode, 'frames_per_gpu': F, 'markers': M, 'free_vars_step1': 3 + len(solver.ids['step1']),
                   'free_vars_step2': 3 + len(solver.ids['step2']), 'sequences_per_gpu': 1},
        'value_is': 'all timed frames / all timed seconds, max over ranks (rounds 1-3 printed the median over seeds of frames / mean step time: `median_over_seeds`)',
        'seeds': per_seed, 'seed_min': round(float(rates.min()), 1), 'seed_max': round(float(rates.max()), 1),
        'median_over_seeds': round(median_rate, 1), 'aggregate_frames_per_s': round(aggregate, 1),
        'timed_mode': f"{args.mode} (mosh_stageii's default chain_mode 'auto' picks '{solver.choose_chain_mode(F)}' for this workload: chmosh.StageIISolver.choose_chain_mode; "
                      f"chain_mode='sequential' -- the run-to-run bit-reproducible choice, and what 'auto' keeps for finger / face / DMPL solves and short captures -- is `sequential_chain` below)",
        'default_mode': f"auto -> {solver.choose_chain_mode(F)}",
    }
    if solver.choose_chain_mode(F) == args.mode:
        result['default_mode_frames_per_s'] = round(value, 1)
    if rep:
        result['chunking'] = dict(rep, repaired_per_step=float(np.mean([r['n_repaired'] for sd in used for r in reports[sd]])))
    if args.mode == 'chunked' and world == 1:
        # the same steps with PLAIN repair chains carried on inside the first launch (rounds 2-3's scheme; MOSHII_COOP_GROUP(1)), once per
        # seed, beside the timed default (cooperative sweeps from the host's rounds where the solve allows)
        tps = {}
        stream_ = torch.cuda.current_stream().cuda_stream
        for sd in used:
            torch.cuda.synchronize()
            tq0 = time.perf_counter()
            seqs[sd].solve_chunked(stream_, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol, coop=1)
            torch.cuda.synchronize()
            tps[str(sd)] = round((time.perf_counter() - tq0) * 1e3, 3)
            seqs[sd].solve_chunked(stream_, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol)   # leave the default's rows
        torch.cuda.synchronize()
        result['plain_sweeps'] = {'ms_per_step_by_seed': tps, 'aggregate_frames_per_s': round(sum(solved_of[sd] for sd in used) / (sum(tps.values()) * 1e-3), 1),
                                  'note': 'one untimed-warm step per seed with one-workgroup repair chains carried on inside the first launch'}

    # ---- fixed jobs at this N (all ranks take part)
    strong = None
    headline_is_strong = (args.scaling == 'strong' or (args.scaling == 'auto' and world > 1)) and args.mode == 'chunked' and not args.no_strong
    if not args.no_strong and args.mode == 'chunked':
        strong = {}
        try:
            n_here, t_here, n_local, t_one, n_one, s_reps, s_warm = strong_many()
            t_job = allmax(t_here)
            n_job = allsum(n_here)
            if dist is not None:    # every rank's own seconds: who waited for whom
                tl = [None] * world
                dist.all_gather_object(tl, float(t_here))
            else:
                tl = [float(t_here)]
            strong['many_sequences'] = {
                'workload': f'{args.strong_sequences} DISTINCT {F}-frame SMPL-H captures of the seed-{seeds[0]} subject (BASELINE config 3 shape with body markers; motion seeds 5000..), dealt to the ranks by '
                            'longest-processing-time (parallel.partition_units); no collective on the data path',
                'frames': int(n_job), 'frames_per_s': round(n_job / t_job, 1), 'ms': round(t_job * 1e3, 2), 'timed_passes': s_reps, 'warmup_passes': s_warm,
                'sequences_on_rank0': n_local, 'rank0_idle_ms': round((t_job - t_here) * 1e3, 2),
                'rank_ms': [round(x * 1e3, 2) for x in tl], 'rank_idle_ms': [round((t_job - x) * 1e3, 2) for x in tl]}
            if t_one:
                strong['many_sequences']['one_gpu_same_job'] = {'frames_per_s': round(n_one / t_one, 1), 'ms': round(t_one * 1e3, 2),
                                                                'speedup': round((n_job / t_job) / (n_one / t_one), 3),
                                                                'note': 'rank 0 alone on the whole job, same box, before the sharded run'}
        except Exception as e:
            strong['many_sequences'] = {'error': repr(e)}
        try:
            n_here, t_here, info, ncalls = strong_long()
            t_job = allmax(t_here)
            n_job = allsum(n_here)
            strong['long_sequence'] = {
                'workload': f'one {args.long_frames}-frame SMPL-H sequence (BASELINE config 5 shape) cut into {world} frame range(s), '
                            f'{args.chunk_warmup}-frame warm-up overlap, hand-offs verified to {args.verify_tol:g} and repaired '
                            '(parallel.solve_sequence_sharded); host buffers (PCIe staging inside the time)',
                'frames': int(n_job), 'frames_per_s': round(n_job / t_job, 1), 'ms': round(t_job * 1e3, 2),
                'handoff_repair_rounds': int(info['rounds']), 'ranks_repaired': info['repaired'],
                'max_handoff_dev': float(info['max_handoff_dev']), 'rank0_idle_ms': round((t_job - t_here) * 1e3, 2),
                'solves_on_rank0': ncalls}
        except Exception as e:
            strong['long_sequence'] = {'error': repr(e)}
        result['strong'] = strong
        if (args.scaling == 'strong' or (args.scaling == 'auto' and world > 1)) and 'frames_per_s' in strong.get('many_sequences', {}):
            sj = strong['many_sequences']
            result.update(value=sj['frames_per_s'], scaling='strong', ms_per_step=sj['ms'], steps=sj['timed_passes'], warmup=sj['warmup_passes'],
                          value_is='job frames / max-over-ranks seconds per pass of the fixed many-sequence job (a step = one pass over the job)')
            result['config'] = {'workload': sj['workload'], 'mode': args.mode, 'markers': M,
                                'parallelism': f'{world} rank(s) = {world} GPU(s), one process each; sequences sharded by longest-processing-time, no data-path collective'}
            result['replicas'] = {'value': round(value, 2), 'scaling': 'weak', 'seeds': per_seed}
            if 'one_gpu_same_job' in sj:   # the same job on ONE GPU of this box (rank 0 alone, before the sharded run): what `value` is to be divided by
                result['one_gpu_same_job'] = sj['one_gpu_same_job']
                result['speedup_vs_one_gpu_same_job'] = sj['one_gpu_same_job']['speedup']

    if rank == 0:
        nobs_mean = float(job['vis'].sum(1).mean())
        fl_seed = {}
        for sd in used:
            st = res[sd]['status'] != 1
            it = res[sd]['iters']
            fl_seed[sd] = solver_flops(sm.K, 3 * M, len(solver.ids['step1']), len(solver.ids['step2']), sm.NP, len(solver.ids['body']),
                                       float(jobs[sd]['vis'].sum(1).mean()), it[st, 0].sum(), it[st, 1].sum())
        fl = fl_seed[seeds[0]]
        fl_timed = sum(fl_seed[seeds[k % len(seeds)]] for k in range(args.steps))
        kt = float(step_ms.sum()) * 1e-3
        ach = fl_timed / kt / 1e12
        pmc, pmc_note = load_pmc(args.pmc_file, capi.load().moshii_source_hash().decode())
        pmc_is = (f'from {os.path.relpath(args.pmc_file, ROOT)} (rocprofv3 PMC passes on this build of the library), NOT collected in this run' if pmc
                  else 'null: no PMC record of this build of the library')
        result['roofline'] = {
            'kernel': name, 'bound': 'valu_f64',
            'bound_note': 'neither hbm nor mfma: float64 vector pipe, instruction-count / latency-bound small dense solves with one wave per SIMD; memory side 45 KB/frame for a one-workgroup chain alone, ~476 KB/frame (mostly scratch write-back) with a chain on every CU, 1.6 MB/frame for a cooperative chain (write-through exchanges) (PMC)',
            'achieved': round(ach, 5), 'peak': F64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / F64_VALU_PEAK_TFLOPS, 6),
            # HBM traffic: PMC passes of tools/r06_collect.sh over one seed-1000 step (--pmc-file), split into the pass-1 launch (bytes per
            # frame it solves: chunks x (chunk + warm-up) frames) and the cooperative repair rounds of that step; null when the file was
            # collected on another build of the library
            'traffic': (int(pmc['chain']['pass1_bytes_per_solved_frame'] * (F + (rep['n_chunks'] * rep['warmup'] if rep else 0)) + (pmc['chain']['repair_bytes_per_step'] if rep else 0))
                        if pmc and 'chain' in pmc else None),
            'traffic_is': pmc_is, 'traffic_source': pmc_note,
            # the same fraction seed by seed (round 1 quoted seed 1000 alone: 0.0058; `frac` above is over all timed steps)
            'frac_by_seed': {str(sd): round(sum(fl_seed[sd] for k in range(args.steps) if seeds[k % len(seeds)] == sd)
                                            / max(sum(float(step_ms[k]) for k in range(args.steps) if seeds[k % len(seeds)] == sd) * 1e-3, 1e-12)
                                            / 1e12 / F64_VALU_PEAK_TFLOPS, 6)
                             for sd in seeds if any(seeds[k % len(seeds)] == sd for k in range(args.steps))},
            'step_ms_hip_events': round(float(step_ms.mean()), 3), 'step_ms_hip_events_all': [round(float(x), 3) for x in step_ms],
            'algorithmic_gflop_per_step': round(fl_timed / args.steps / 1e9, 3),
            'note': 'algorithmic = the sequential chain\'s work on the recorded frames of the timed steps / their HIP-event time; warm-up and repair work is overhead',
            'dogleg_iters_per_frame': round(float(iters[solved_mask, 0].sum()) / max(solved, 1), 3),
            'residual_evals_per_frame': round(float(iters[solved_mask, 1].sum()) / max(solved, 1), 3), 'lds_bytes': lds,
        }
        extras = world == 1     # the reference / CPU / Stage-I legs run at N = 1 only: at N > 1 the other ranks would idle at the barrier
        # ---- the literal frame order on one workgroup, and the timed mode's deviation from it over all frames
        if extras and args.mode == 'chunked' and not args.no_sequential:
            dsq = workload.DeviceSequence(job, solver, dev)
            stream = torch.cuda.current_stream().cuda_stream
            # the ONE-workgroup chain first (rounds 1-3's `sequential_chain`), then the chain as the library runs it when asked for the
            # sequential order: a cooperative chain of several workgroups where the solve allows
            dsq.solve_sequential(stream, coop=1)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            dsq.solve_sequential(stream, coop=1)
            torch.cuda.synchronize()
            tplain = time.perf_counter() - tp0
            kplain = capi.last_launch_info()[0]
            splain = dsq.results()
            dsq.solve_sequential(stream)
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            dsq.solve_sequential(stream)
            torch.cuda.synchronize()
            tsq = time.perf_counter() - ts0
            sq = dsq.results()
            dp = np.abs(out['fullpose'] - sq['fullpose'])[solved_mask].max(1)
            dm = out['markers_sim'][solved_mask] - sq['markers_sim'][solved_mask]
            result['sequential_chain'] = {
                'frames_per_s': round(solved / tsq, 1), 'ms': round(tsq * 1e3, 1), 'us_per_frame': round(tsq * 1e6 / max(solved, 1), 1),
                'kernel': capi.last_launch_info()[0], 'seed': seeds[0],
                'one_workgroup': {'frames_per_s': round(solved / tplain, 1), 'us_per_frame': round(tplain * 1e6 / max(solved, 1), 1), 'kernel': kplain,
                                  'max_abs_pose_diff_vs_default_rad': float(np.abs(splain['fullpose'] - sq['fullpose'])[solved_mask].max()),
                                  'iteration_counts_identical': bool((splain['iters'] == sq['iters']).all())},
                'timed_mode_vs_sequential': {'max_abs_pose_diff_rad': float(dp.max()), 'frames_over_1e-4_rad': int((dp > 1e-4).sum()),
                                             'frames_over_1e-6_rad': int((dp > 1e-6).sum()),
                                             'marker_rmse_m': float(np.sqrt((dm ** 2).sum(-1).mean())),
                                             'status_identical': bool((status == sq['status']).all())}}
            result['speedup_vs_sequential_chain'] = round(value / max(solved / tsq, 1e-9), 2)
            if solver.choose_chain_mode(F) == 'sequential':
                result['default_mode_frames_per_s'] = round(solved / tsq, 1)
            del dsq
            # ... and over ALL timed seeds: the timed (chunked) result of every seed against that seed's own sequential chain -- frames
            # over the north-star 1e-4 rad (ill-conditioned stretches: a 1e-13 hand-off difference amplified to another local solution,
            # DESIGN.md section 3) and the worst per-frame marker RMSE between the two results and against the observations
            per, env = {}, {}
            for sd in used:
                dq = workload.DeviceSequence(jobs[sd], solvers[sd], dev)
                dq.solve_sequential(stream)
                torch.cuda.synchronize()
                rs, rc = dq.results(), res[sd]
                ok = rc['status'] != 1
                dps = np.abs(rc['fullpose'] - rs['fullpose'])[ok].max(1)
                vis_s = jobs[sd]['vis'][ok]
                fr = np.sqrt((((rc['markers_sim'] - rs['markers_sim'])[ok] ** 2).sum(-1) * vis_s).sum(1) / np.maximum(vis_s.sum(1), 1))
                fo = np.sqrt((((rc['markers_sim'] - jobs[sd]['obs'])[ok] ** 2).sum(-1) * vis_s).sum(1) / np.maximum(vis_s.sum(1), 1))
                per[str(sd)] = {'max_abs_pose_diff_rad': float(dps.max()), 'frames_over_1e-4_rad': int((dps > 1e-4).sum()),
                                'worst_frame_marker_rmse_vs_sequential_m': float(fr.max()), 'worst_frame_marker_rmse_vs_observations_m': float(fo.max()),
                                'status_identical': bool((rc['status'] == rs['status']).all())}
                # EVERY frame of BOTH modes against the committed ORACLE trajectory of the seed and its sensitivity envelope
                # (tests/parity_envelope.py; tests/golden/oracle_traj_seed<seed>.npz: the NumPy oracle over the whole sequence + K perturbed
                # oracle runs) -- fixtures, read as the checker; they exist for the default workload (4000 frames, 53 markers)
                try:
                    from tests import parity_envelope as pe
                    if pe.have(sd) and F == 4000 and M == 53:
                        solved_ids = np.flatnonzero(rs['status'] == 0)
                        mo, _, clo, cof = oracle_setup(jobs[sd])
                        vv = jobs[sd]['vis'][solved_ids]
                        e_def = pe.check(sd, rs['pose'][solved_ids], rs['trans'][solved_ids], rs['iters'][solved_ids], frames=solved_ids,
                                         markers_sim=rs['markers_sim'][solved_ids], vis=vv, oracle_model=(mo, clo, cof))
                        e_tim = pe.check(sd, rc['pose'][solved_ids], rc['trans'][solved_ids], frames=solved_ids,
                                         markers_sim=rc['markers_sim'][solved_ids], vis=vv, oracle_model=(mo, clo, cof))
                        env[str(sd)] = {'sequential_chain': dict(e_def, ok=pe.ok(e_def)), 'timed_mode_chunked': dict(e_tim, ok=pe.ok(e_tim))}
                except Exception as e:
                    env[str(sd)] = {'error': repr(e)}
                del dq
            result['sequential_chain']['timed_mode_vs_sequential_all_seeds'] = per
            if env:
                good = [v for v in env.values() if 'error' not in v]
                both = [v[k] for v in good for k in ('sequential_chain', 'timed_mode_chunked')]
                result['parity_every_frame'] = {
                    'against': 'the NumPy oracle over the WHOLE sequence of every timed seed (tests/golden/oracle_traj_seed*.npz, made by tests/golden/make_oracle_trajectories.py); '
                               'configs 3 / 4 / 5 at their stated lengths under `configs` (make_oracle_trajectories_configs.py)',
                    'criterion': 'tests/parity_envelope.py: |state - oracle| <= 1e-7, equal dogleg iteration counts and <= 1e-6 m marker RMSE per frame wherever 3 oracle runs on observations '
                                 'perturbed by 1e-13 m stay within 3e-9 of the oracle; where they part (a knife edge of the reference algorithm itself) a trajectory may part too, by at most '
                                 'max(1e-3, 30 x the stretch\'s spread), and has to be back within 64 frames of the stretch\'s end; whole-sequence marker RMSE against the ORACLE\'s simulated markers <= 1e-3 m',
                    'frames_checked': int(sum(v['frames'] for v in both)),
                    'frames_outside_tolerance': int(sum(v['frames_outside_tolerance'] for v in both)),
                    'frames_parted_on_a_knife_edge': int(sum(v['frames_parted_on_a_knife_edge'] for v in both)),
                    'frames_over_1e-4_rad': int(sum(v['frames_over_1e-4_rad'] for v in both)),
                    'max_dev_on_well_conditioned_frames_rad': float(max([v['max_dev_on_well_conditioned_frames_rad'] for v in both] or [0.0])),
                    'worst_sequence_marker_rmse_vs_oracle_m': float(max([v['marker_rmse_vs_oracle_m'] for v in both] or [0.0])),
                    'worst_frame_marker_rmse_vs_oracle_m': float(max([v['worst_frame_marker_rmse_vs_oracle_m'] for v in both] or [0.0])),
                    'all_ok': bool(all(v['ok'] for v in both)),
                    'by_seed': env}
            result['sequential_chain']['frames_over_1e-4_rad_all_seeds'] = int(sum(v['frames_over_1e-4_rad'] for v in per.values()))
            result['sequential_chain']['worst_frame_marker_rmse_vs_sequential_m_all_seeds'] = float(max(v['worst_frame_marker_rmse_vs_sequential_m'] for v in per.values()))
        # ---- the same solve through host buffers (PCIe staging of observations and results inside the time)
        if extras and args.mode == 'chunked':
            try:
                solver.solve(job['obs'][:64], job['vis'][:64], chain_mode='chunked', verify_tol=args.verify_tol)
                th0 = time.perf_counter()
                oh = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=args.verify_tol)
                th = time.perf_counter() - th0
                result['incl_host_staging'] = {'frames_per_s': round(int((oh['status'] != 1).sum()) / th, 1), 'ms': round(th * 1e3, 2), 'seed': seeds[0],
                                               'note': 'moshii_sequence_solve with MOSHII_BUFFERS_HOST: pageable host arrays in and out'}
            except Exception as e:
                result['incl_host_staging'] = {'error': repr(e)}
        # ---- the same kernel with the chip full = the N = 1 point of strong.many_sequences
        if extras and strong and 'frames_per_s' in strong.get('many_sequences', {}):
            sj = strong['many_sequences']
            fl_many = fl * args.strong_sequences
            mach = fl_many / (sj['ms'] * 1e-3) / 1e12
            result['many_sequences'] = {'sequences': args.strong_sequences, 'frames': sj['frames'], 'frames_per_s': sj['frames_per_s'], 'ms': sj['ms'],
                                        'roofline': {'kernel': 'k_chain_solve<4,1> chunk chains + cooperative repair chains', 'bound': 'valu_f64', 'achieved': round(mach, 4), 'peak': F64_VALU_PEAK_TFLOPS,
                                                     'unit': 'TFLOP/s', 'frac': round(mach / F64_VALU_PEAK_TFLOPS, 5)}}
        # ---- BASELINE config 3 as stated: 32 x 4000-frame SMPL-X captures, 89 markers incl. face / hands, fingers + jaw + 80 expression
        # coefficients free (194 unknowns per Step-2 solve; chmosh.py:560-567, 681-689).  With a free expression block a chunk start
        # never reproduces the chain's coefficients, so this size class runs one sequential chain per sequence:
        # 32 of the 256 CUs carry a chain.  Host arrays in and out (the extended kernel's buffers), staging included.
        if extras and not args.no_config3:
            try:
                fj = workload.make_face_job()
                fsol = workload.make_solver(fj)
                caps = [workload.make_face_capture(fj, fsol, 7000 + i, n_frames=args.config3_frames) for i in range(args.config3_sequences)]
                chains = [dict(attach=fsol.attach, obs=c['obs'], vis=c['vis'], first=True) for c in caps]
                capi.chain_solve_host(fsol.dev, fsol.prior, fsol.opts, [dict(attach=fsol.attach, obs=caps[0]['obs'][:8], vis=caps[0]['vis'][:8], first=True)])
                t0 = time.perf_counter()
                fouts = capi.chain_solve_host(fsol.dev, fsol.prior, fsol.opts, chains)
                dt3 = time.perf_counter() - t0
                kname3 = capi.last_launch_info()[0]
                nfr = sum(int((o['status'] != 1).sum()) for o in fouts)
                it3 = np.concatenate([o['iters'][o['status'] != 1] for o in fouts])
                d3 = np.concatenate([(o['markers_sim'] - c['obs'])[c['vis']] for o, c in zip(fouts, caps)])
                ex3 = max(float(np.abs(o['shape'][-1] - c['expr_gt']).max()) for o, c in zip(fouts, caps))
                smx = fj['sm']
                fl3 = solver_flops(smx.K, 3 * caps[0]['vis'].shape[1], len(fsol.ids['step1']), len(fsol.ids['step2']) + fsol.n_shape, smx.NP,
                                   len(fsol.ids['body']), float(np.mean([c['vis'].sum(1).mean() for c in caps])), it3[:, 0].sum(), it3[:, 1].sum())
                result['config3'] = {
                    'workload': f'BASELINE config[2] as stated: {args.config3_sequences} distinct {args.config3_frames}-frame SMPL-X captures of one subject, '
                                f'{caps[0]["vis"].shape[1]} markers incl. face / hand vertices, fingers + jaw + {fsol.n_shape} expression coefficients free '
                                f'({3 + len(fsol.ids["step2"]) + fsol.n_shape} unknowns per Step-2 solve); one sequential chain per sequence in ONE launch '
                                f'({args.config3_sequences} of the CUs busy); host arrays in and out',
                    'kernel': kname3, 'frames': nfr, 'frames_per_s': round(nfr / dt3, 1), 'seconds': round(dt3, 2),
                    'us_per_frame_per_chain': round(1e6 * dt3 / args.config3_frames, 1),
                    'dogleg_iterations_per_frame': round(float(it3[:, 0].mean()), 2),
                    'marker_rmse_m': float(np.sqrt((d3 ** 2).sum(1).mean())), 'final_expression_max_err': ex3,
                    'roofline': {'kernel': kname3, 'bound': 'valu_f64', 'achieved': round(fl3 / dt3 / 1e12, 4), 'peak': F64_VALU_PEAK_TFLOPS,
                                 'unit': 'TFLOP/s', 'frac': round(fl3 / dt3 / 1e12 / F64_VALU_PEAK_TFLOPS, 5)}}
            except Exception as e:
                result['config3'] = {'error': repr(e)}
        # ---- BASELINE configs 3, 4, 5 at their stated lengths against the committed ORACLE trajectories + envelopes (the GPU tier's
        #      test_config{3,4,5}_*_against_the_oracle; here as numbers on the line): captures 7000 and 7001 of the config-3 subject (4000 frames, 194
        #      unknowns), one MANO hand (10 000 frames), the first 8000 frames of the 50 000-frame config-5 capture
        if extras and not args.no_cpu and 'parity_every_frame' in result:
            cfgs = {}
            try:
                from tests import parity_envelope as pe
                from tests.golden.make_oracle_trajectories_configs import case_inputs

                def envelope(case, o, vis, om, shape=False, iters=True):
                    sol = np.flatnonzero(o['status'] <= 0)
                    sol = sol[sol <= pe.load(case)['frame_ids'][-1]]
                    r = pe.check(case, o['pose'][sol], o['trans'][sol], o['iters'][sol] if iters else None, frames=sol, shape=o['shape'][sol] if shape else None,
                                 markers_sim=o['markers_sim'][sol], vis=vis[sol], oracle_model=om)
                    return dict({k: r[k] for k in ('frames', 'frames_outside_tolerance', 'frames_parted_on_a_knife_edge', 'frames_over_1e-4_rad',
                                                    'max_dev_on_well_conditioned_frames_rad', 'max_dev_on_parted_frames_rad', 'max_oracle_spread_rad',
                                                    'marker_rmse_vs_oracle_m', 'worst_frame_marker_rmse_vs_oracle_m')}, ok=pe.ok(r))
                if not args.no_config3:
                    for cap3 in (7000, 7001):   # (7000: the perturbed oracle runs part on 628 frames, by up to 0.68 rad; 7001: well conditioned throughout)
                        c3 = case_inputs(f'config3_{cap3}')
                        s3 = workload.make_solver(c3['job'])
                        o3 = capi.chain_solve_host(s3.dev, s3.prior, s3.opts, [dict(attach=s3.attach, obs=c3['obs'], vis=c3['vis'], first=True)])[0]
                        cfgs[f'config3_capture_{cap3}_4000_frames'] = envelope(f'config3_{cap3}', o3, c3['vis'], (c3['m'], c3['closest'], c3['coef']), shape=True)
                        del s3
                c4 = case_inputs('mano_72')
                s4 = workload.make_solver(c4['job'])
                o4 = s4.solve(c4['job']['obs'], c4['job']['vis'], chain_mode='sequential')
                cfgs['config4_mano_hand_10000_frames'] = envelope('mano_72', o4, c4['job']['vis'], (c4['m'], c4['closest'], c4['coef']))
                del s4
                c5 = case_inputs('config5_1000')
                s5 = workload.make_solver(c5['job'])
                o5 = s5.solve(c5['job']['obs'], c5['job']['vis'], chain_mode='auto', verify_tol=args.verify_tol)
                cfgs['config5_first_8000_of_50000_frames'] = dict(envelope('config5_1000', o5, c5['job']['vis'], (c5['m'], c5['closest'], c5['coef']), iters=False), mode=o5.get('chain_mode'))
                del s5
            except Exception as e:
                cfgs['error'] = repr(e)
            pv = result['parity_every_frame']
            pv['configs'] = cfgs
            goodc = [v for v in cfgs.values() if isinstance(v, dict)]
            pv['frames_checked'] += int(sum(v['frames'] for v in goodc))
            pv['frames_outside_tolerance'] += int(sum(v['frames_outside_tolerance'] for v in goodc))
            pv['frames_parted_on_a_knife_edge'] += int(sum(v['frames_parted_on_a_knife_edge'] for v in goodc))
            pv['frames_over_1e-4_rad'] += int(sum(v['frames_over_1e-4_rad'] for v in goodc))
            pv['all_ok'] = bool(pv['all_ok'] and all(v['ok'] for v in goodc) and 'error' not in cfgs)
        # ---- full-mesh LBS export kernel (the kernel the HBM-roofline target names).  Two bodies of the same sizes: the synthetic SMPL-H
        #      with its vertices in MESH order (bone by bone, along each bone -- how a registered artist mesh numbers them: consecutive
        #      ids share joints, which is what the kernel's per-group joint lists profit from) -- `roofline_lbs` -- and the same body
        #      with SHUFFLED vertex ids (the worst case; what every solver fixture of this repository uses) beside it.
        try:
            import ctypes as C
            from moshpp_amd import synth
            Fl = min(args.lbs_frames, F)
            pose32 = ds.pose[:Fl].to(torch.float32).contiguous()
            trans32 = ds.trans[:Fl].to(torch.float32).contiguous()
            stream = torch.cuda.current_stream().cuda_stream

            def lbs_leg(slv, tag, pose_t, what, pmc_tag=None):
                smv = slv.sm if hasattr(slv, 'sm') else sm
                verts = torch.empty((Fl, smv.V, 3), dtype=torch.float32, device=dev)
                for _ in range(2):
                    slv.dev.lbs_forward_device(Fl, pose_t.data_ptr(), trans32.data_ptr(), verts.data_ptr(), C.c_void_p(stream))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    slv.dev.lbs_forward_device(Fl, pose_t.data_ptr(), trans32.data_ptr(), verts.data_ptr(), C.c_void_p(stream))
                e1.record()
                torch.cuda.synchronize()
                lt = e0.elapsed_time(e1) * 1e-3 / reps
                # a 60-frame spot check of the timed output against the reference-precision kernel
                chk = slv.dev.lbs_forward(pose_t[:60].cpu().numpy().astype(np.float64), trans32[:60].cpu().numpy().astype(np.float64))
                err = float(np.abs(verts[:60].cpu().numpy() - chk).max())
                Kj = smv.K
                # algorithmic bytes as SURVEY.md section 8(d) / BASELINE.md section 4 state them: 12 V out + pose/trans in per frame, plus ONE
                # read of the f32 model, 12 V (1 + 9 (K - 1)) + 4 V K bytes (SMPL-H: 39.5 MB; the kernel actually reads f16 posedirs -- 19 MB:
                # `frac_with_f16_model_bytes` is the same time under rounds 3-4's accounting)
                model_bytes = 12 * smv.V * (1 + 9 * (Kj - 1)) + 4 * smv.V * Kj
                bytes_alg = Fl * (12 * smv.V + 4 * smv.NP + 12) + model_bytes
                bytes_f16 = Fl * (12 * smv.V + 4 * smv.NP + 12) + 6 * smv.V * 9 * (Kj - 1) + 16 * smv.V
                pm = pmc.get('lbs', {}).get(pmc_tag or tag) if pmc else None
                return {'kernel': 'k_lbs_export (+ k_lbs_prep, k_lbs_still)', 'bound': 'hbm', 'body': tag, 'poses': what,
                        'dtype': 'f32 out; f16-operand / f32-accumulate MFMA correctives, f32 blend on the f32 matrix instruction',
                        'achieved': round(bytes_alg / lt / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(bytes_alg / lt / 1e9 / HBM_PEAK_GBS, 4),
                        'frac_with_f16_model_bytes': round(bytes_f16 / lt / 1e9 / HBM_PEAK_GBS, 4),
                        'algorithmic_bytes': int(bytes_alg),
                        'traffic': int(pm['bytes_per_call_at_4000_frames'] * Fl / 4000.0) if pm else None, 'traffic_is': pmc_is, 'traffic_source': pmc_note,
                        'frames': Fl, 'kernel_ms': round(lt * 1e3, 3), 'frames_per_s': round(Fl / lt, 1),
                        'max_abs_err_vs_f64_kernel_m_first_60_frames': err}

            # the poses: (a) the SOLVED sequence as it is -- a body-only solve (the reference's default, optimize_fingers off) leaves the hand
            # pose at the prior's mean in every frame, which the export notices per call (k_lbs_prep's marks) and folds into the rest
            # positions (k_lbs_still): 6 of SMPL-H's 15 k-steps remain; (b) the same sequence with every hand-pose variable moving
            what_a = 'the solved config[1] sequence (body-only Stage-II: the hand pose is the same in every frame)'
            what_b = 'the solved sequence with seeded N(0, 0.3) hand-pose variables in every frame (every joint moves)'
            pose_mv = pose32.clone()
            gen = torch.Generator(device='cpu'); gen.manual_seed(1234)
            pose_mv[:, sm.body_dof:] = (0.3 * torch.randn((Fl, sm.NP - sm.body_dof), generator=gen)).to(dev)
            dd_mesh = synth.synth_model('smplh', seed=seeds[0], vertex_order='mesh')
            job_mesh = workload.make_job('smplh', n_frames=8, n_markers=M, seed=seeds[0], dd=dd_mesh)
            solver_mesh = workload.make_solver(job_mesh)
            result['roofline_lbs'] = lbs_leg(solver_mesh, 'mesh_order', pose32, what_a)
            result['roofline_lbs']['all_joints_moving'] = lbs_leg(solver_mesh, 'mesh_order', pose_mv, what_b, 'mesh_order_moving')
            result['roofline_lbs']['shuffled_vertex_ids'] = lbs_leg(solver, 'shuffled_ids', pose32, what_a)
            result['roofline_lbs']['shuffled_vertex_ids_all_joints_moving'] = lbs_leg(solver, 'shuffled_ids', pose_mv, what_b, 'shuffled_ids_moving')
            result['roofline_lbs']['limit_note'] = (
                'round 6: the blend runs on v_mfma_f32_16x16x4_f32 (12 matrix instructions a 16-frame block and round of four joints; the packed FMAs of round 5 '
                'could not overlap the matrix pipe at all: profiles/r06_ubench_valu.txt), still joints leave the k-loop; what remains per 64 x 128 tile and wave is '
                '~6 k-steps x 24 + 96 f32 matrix instructions + ~80 vector / LDS / memory instructions a block, issue-bound at two waves per SIMD, plus ~25 us of '
                'k_lbs_prep + k_lbs_still + kernel boundaries per call (DESIGN.md section 6)')
            del solver_mesh
        except Exception as e:   # the LBS leg must never take the headline number down
            result['roofline_lbs'] = {'error': repr(e)}
        # ---- CPU baseline: the NumPy oracle ("port") on bounded samples of the same workload; parity on the first sample
        if extras and not args.no_cpu:
            from oracle import stageii_oracle as so
            S = min(args.cpu_sample, F)
            m, pr, closest, coef = oracle_setup(job)
            # one core: the chain is sequential and its matrices are small (<= 385 x 111) -- BLAS threading buys nothing
            # (measured: 23.8 vs 23.5 frames/s with 8 vs 1 threads), so the library is pinned to one thread and says so
            import contextlib
            try:
                import threadpoolctl
                pin, blas_threads = threadpoolctl.threadpool_limits(limits=1), 1
            except Exception:
                pin, blas_threads = contextlib.nullcontext(), os.cpu_count()
            with pin:
                tc0 = time.perf_counter()
                ref = so.stageii_chain(m, pr, closest, coef, job['obs'][:S], job['vis'][:S], 'smplh')
                tc = time.perf_counter() - tc0
                # the reference's own cost per iteration: full-mesh forward and the dense 3V x 3K vertex Jacobian, marker rows taken
                # from it (smpl_fast_derivatives.py:250-256) -- same trajectory, a bounded number of frames
                Sr = min(6, S)
                tr0 = time.perf_counter()
                refc = so.stageii_chain(m, pr, closest, coef, job['obs'][:Sr], job['vis'][:Sr], 'smplh', reference_cost=True)
                trc = time.perf_counter() - tr0
            n_ref = len(ref['frame_ids'])
            cb = {'value': round(n_ref / tc, 2), 'unit': 'frames/s', 'cores': int(blas_threads), 'kind': 'port',
                  'host_cores_visible': os.cpu_count(),
                  'sample': f'first {S} frames of the seed-{seeds[0]} sequence, NumPy float64 oracle (lean marker-subset mode), single '
                            f'process, BLAS pinned to {blas_threads} thread(s), {tc:.1f} s',
                  'reference_cost': {'value': round(len(refc['frame_ids']) / trc, 3), 'unit': 'frames/s', 'cores': int(blas_threads),
                                     'sample': f'first {Sr} frames, full-mesh forward + dense 3V x 3K Jacobian per iteration as the '
                                               f'reference computes them (smpl_fast_derivatives.py:250-256), {trc:.1f} s',
                                     'max_abs_pose_diff_vs_lean_rad': float(np.abs(refc['fullpose'] - ref['fullpose'][:len(refc['fullpose'])]).max())}}
            try:    # how the reference is deployed: one capture per OS process (mosh_head.py:584-589) -- one sequence per core
                import multiprocessing as mp
                P = args.cpu_procs or min(os.cpu_count() or 1, 32)
                fr = 60
                ctx = mp.get_context('spawn')
                ta0 = time.perf_counter()
                with ctx.Pool(P) as pool:
                    rs = pool.map(_cpu_chain_worker, [(seeds[i % len(seeds)], fr, M) for i in range(P)])
                ta = time.perf_counter() - ta0
                ac = sum(n for n, _ in rs) / max(t for _, t in rs)
                host_threads = os.cpu_count() or P
                cb['all_cores'] = {'value': round(ac, 1), 'unit': 'frames/s', 'cores': P, 'host_threads': host_threads,
                                   # what the whole host would do if the rate per process held on all of its threads (it will not quite:
                                   # SMT siblings, memory bandwidth) -- the honest box-level figure to hold an 8-GPU node against
                                   'extrapolated_to_all_host_threads': round(ac * host_threads / P, 1),
                                   'sample': f'{P} processes x {fr} frames (one sequence per core, lean mode), slowest chain '
                                             f'{max(t for _, t in rs):.1f} s, {ta:.1f} s incl. process start-up'}
            except Exception as e:
                cb['all_cores'] = {'error': repr(e)}
            result['cpu_baseline'] = cb
            gp = out['fullpose'][:S][status[:S] != 1]
            gm = out['markers_sim'][:S]
            sqd = []
            for i, t in enumerate(ref['frame_ids']):
                sqd.append(((gm[t][job['vis'][t]] - ref['markers_sim'][i]) ** 2).sum(1))
            dpo = np.abs(gp - ref['fullpose']).max(1)
            result['parity'] = {'against': 'oracle sequential chain', 'frames': int(n_ref), 'seed': seeds[0],
                                'max_abs_pose_diff_rad': float(dpo.max()), 'frames_over_1e-4_rad': int((dpo > 1e-4).sum()),
                                'marker_rmse_m': float(np.sqrt(np.concatenate(sqd).mean())),
                                'tolerance': {'pose_rad': 1e-4, 'marker_rmse_m': 1e-3}}
            # ... and EVERY timed seed against the oracle on its own first S frames (one oracle process per seed, side by side):
            # where the 1e-4 rad claim holds on the line itself, not only on seed 1000
            try:
                import multiprocessing as mp
                others = [sd for sd in used if sd != seeds[0]]
                by_seed = {str(seeds[0]): {'frames': int(n_ref), 'max_abs_pose_diff_rad': float(dpo.max()),
                                           'frames_outside_tolerance_vs_oracle': int((dpo > 1e-4).sum()),
                                           'marker_rmse_m': float(np.sqrt(np.concatenate(sqd).mean()))}}
                if others:
                    with mp.get_context('spawn').Pool(min(len(others), os.cpu_count() or 1)) as pool:
                        refs = pool.map(_cpu_parity_worker, [(sd, F, M, S) for sd in others])
                    for sd, fids, rfp, rms in refs:
                        g = res[sd]
                        dq_ = np.abs(g['fullpose'][fids] - rfp).max(1)
                        sq_ = np.concatenate([((g['markers_sim'][t][jobs[sd]['vis'][t]] - rms[i]) ** 2).sum(1) for i, t in enumerate(fids)])
                        by_seed[str(sd)] = {'frames': int(len(fids)), 'max_abs_pose_diff_rad': float(dq_.max()),
                                            'frames_outside_tolerance_vs_oracle': int((dq_ > 1e-4).sum()),
                                            'marker_rmse_m': float(np.sqrt(sq_.mean()))}
                result['parity']['by_seed'] = by_seed
                result['parity']['frames_outside_tolerance_vs_oracle_all_seeds'] = int(sum(v['frames_outside_tolerance_vs_oracle'] for v in by_seed.values()))
                result['parity']['note'] = (f'the oracle run live on the first {S} frames of every timed seed (the cpu_baseline leg); the FULL-LENGTH check of both '
                                            'modes against the committed oracle trajectories of all timed seeds is `parity_every_frame`')
            except Exception as e:
                result['parity']['by_seed'] = {'error': repr(e)}
            result['speedup_vs_cpu_port'] = round(value / max(n_ref / tc, 1e-9), 1)
            if 'value' in cb.get('all_cores', {}):     # one GPU against the whole host (all threads, extrapolated), beside the one-core ratio
                result['speedup_vs_cpu_all_host_threads'] = round(value / max(cb['all_cores']['extrapolated_to_all_host_threads'], 1e-9), 2)
            result['speedup_vs_cpu_reference_cost'] = round(value / max(len(refc['frame_ids']) / trc, 1e-9), 1)
        # ---- Stage-I leg (BASELINE config 4's calibration part): 12 picked frames, 53 markers, 10 betas on a
        #      triangulated SMPL-H-sized body; the joint solve on the GPU beside the NumPy oracle on the host
        if extras and not args.no_stagei:
            try:
                pb1, dev1, pr1, kw1 = workload.make_stagei_job()
                capi.stagei_solve_host(dev1, pr1, **kw1)
                ts = []
                for _ in range(3):
                    t1 = time.perf_counter(); o1 = capi.stagei_solve_host(dev1, pr1, **kw1); ts.append(time.perf_counter() - t1)
                leg = {'workload': f"12 frames, 53 markers, 10 betas, V={pb1['model']['v_template'].shape[0]}, {len(pb1['faces'])} triangles",
                       'unknowns': int(3 * 12 + 3 * 53 + 12 * len(kw1['pose_ids']) + 10), 'seconds': round(float(np.median(ts)), 4),
                       'dogleg_iterations': o1['iters']}
                leg['solver'] = 'arrow-structured (per-frame elimination + Schur complement on the shared block): the default'
                try:    # the dense blocked Cholesky of the whole system beside it (MOSHII_S1_SOLVER=dense)
                    os.environ['MOSHII_S1_SOLVER'] = 'dense'
                    capi.stagei_solve_host(dev1, pr1, **kw1)
                    t1 = time.perf_counter(); o2 = capi.stagei_solve_host(dev1, pr1, **kw1); leg['seconds_dense_solver'] = round(time.perf_counter() - t1, 4)
                    leg['schur_vs_dense_max_abs_betas_diff'] = float(np.abs(o2['betas'] - o1['betas']).max())
                except Exception as e:
                    leg['seconds_dense_solver'] = repr(e)
                finally:
                    os.environ.pop('MOSHII_S1_SOLVER', None)
                if not args.no_cpu:     # the CPU side of this leg: the NumPy oracle on the same problem (checker + timing)
                    from oracle import stageii_oracle as so1, stagei_oracle as s1o
                    m1 = so1.prepare_model(pb1['model'])
                    so1.set_free_shape(m1, 0, pb1['nb'])
                    t1 = time.perf_counter()
                    r1 = s1o.stagei_solve(m1, pb1['faces'], so1.prepare_gmm_prior(pb1['gmm'], 63), 'smplh', pb1['frames'], pb1['vids'],
                                          {'body': np.ones(pb1['M'], bool)}, {'body': pb1['skin']}, pb1['nb'])
                    leg['cpu_oracle_seconds'] = round(time.perf_counter() - t1, 2)
                    leg['max_abs_betas_diff'] = float(np.abs(o1['betas'] - r1['betas']).max())
                    leg['max_abs_markers_latent_diff_m'] = float(np.abs(o1['markers_latent'] - r1['markers_latent']).max())
                result['stagei'] = leg
            except Exception as e:
                result['stagei'] = {'error': repr(e)}
        result = _finite(result)
        result['detail_file'] = write_detail(result) or 'stderr only'
        line = compact_line(result)
        sys.stdout.flush()
        os.dup2(_real_stdout, 1)           # the ONE line on stdout: <= 4 KB (the whole result: bench_detail.json / stderr)
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
