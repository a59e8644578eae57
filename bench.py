#!/usr/bin/env python
"""Stage-II throughput bench (BASELINE.json metric: solved mocap frames/sec).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full Stage-II pass over one synthetic sequence of BASELINE config[1] (4000-frame SMPL-H, 53 body
markers, fixed betas) with observations and outputs resident in HBM: `moshii_sequence_solve` cuts the sequence into
one chunk per CU, solves the chunks concurrently (each starts --chunk-warmup frames early), verifies every hand-off
against its predecessor's end state on the device and re-solves the chunks that miss --verify-tol exactly -- all of
that is inside the timed region.  `--mode sequential` times the reference's literal frame order instead (one chain =
one workgroup).  With N > 1 (torch.distributed.run, one rank per GPU) every rank solves its own sequence of the same
shape -- the path has no data-path collective -- and the job-level frames/s is reported ("scaling": "weak").

Rank 0 prints ONE JSON line.  Extra objects:
  roofline           dominant kernel k_chain_solve against the f64 vector peak (it is latency/ALU-bound, not HBM-bound)
  roofline_lbs       the full-mesh LBS export kernel against the HBM peak
  cpu_baseline       the NumPy oracle ("port") on a bounded sample, host cores of this box
  parity             GPU result of the timed mode vs the oracle's sequential chain on that sample
  sequential_chain   the literal frame order on the GPU (one workgroup) and the timed mode's deviation from it over
                     ALL frames
  many_sequences     32 copies of the sequence in one call: the dominant kernel with every CU busy
  other_seeds        the same workload generated from two other seeds (the repair pattern depends on the motion)
  stagei             Stage-I on 12 frames / 53 markers / 10 betas: GPU seconds (default dense solver and the opt-in
                     arrow-structured one), iterations, and the NumPy oracle's seconds + differences on the same problem
All extra legs except roofline / roofline_lbs run at N = 1 only.
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


def solver_flops(K, Nv, n1, n2, NP, npose, nobs_mean, iters, fevals):
    """Algorithmic FLOPs of the solve (SURVEY.md 8(d) / DESIGN.md 5): forward 2*3Nv*9(K-1) + 2*Nv*K*12 per residual
    evaluation; per Jacobian 3Nv*n*30 + normal equations 2*m*n^2/2 + factorisation n^3/3, m = 3*nobs + (npose+1) + NP."""
    fwd = 2.0 * 3 * Nv * 9 * (K - 1) + 2.0 * Nv * K * 12
    m = 3.0 * nobs_mean + (npose + 1 if npose else 0) + NP
    n = 3 + 0.5 * (n1 + n2)
    per_iter = 3.0 * Nv * n * 30 + m * n * n + n ** 3 / 3.0
    return float(fevals) * fwd + float(iters) * per_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=4000)
    ap.add_argument('--markers', type=int, default=53)
    ap.add_argument('--mode', choices=('chunked', 'sequential'), default='chunked')
    ap.add_argument('--chunks', type=int, default=0, help='chunks per sequence (0 = one per CU)')
    ap.add_argument('--chunk-warmup', type=int, default=32)
    # hand-off tolerance: 1e-9 rad / m yields the same stitched result as 1e-11 (max deviation from the sequential chain 1.4e-9
    # rad on all 4000 frames, re-measured below every run) with fewer repairs of warm-ups that were converged to 1e-10
    ap.add_argument('--verify-tol', type=float, default=1e-9)
    ap.add_argument('--cpu-sample', type=int, default=400, help='frames of the workload timed on the CPU oracle')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-stagei', action='store_true', help='skip the Stage-I leg')
    ap.add_argument('--no-sequential', action='store_true', help='skip the one-workgroup sequential reference run')
    ap.add_argument('--spread-seeds', default='5,71', help='extra leg: the same workload generated from these seeds ("" to skip)')
    ap.add_argument('--lbs-frames', type=int, default=4000)   # the whole solved sequence: that is what a mesh export writes
    ap.add_argument('--many', type=int, default=32, help='extra leg: this many copies of the sequence in one call (0: skip)')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the Stage-II path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from moshpp_amd import capi, workload
    lib = capi.load()
    capi.check(lib.moshii_set_device(local_rank))

    # ---- workload: BASELINE config[1]; every rank solves its own copy of the same seeded sequence (identical work per
    # GPU: the chunk-repair pattern depends on the motion, so different seeds would blur the weak-scaling figure)
    job = workload.make_job('smplh', n_frames=args.frames, n_markers=args.markers, seed=1000)
    solver = workload.make_solver(job)
    sm = job['sm']
    F, M = job['vis'].shape
    dev = torch.device('cuda', local_rank)
    ds = workload.DeviceSequence(job, solver, dev)
    reports = []

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        if args.mode == 'chunked':
            reports.append(ds.solve_chunked(stream, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol))
        else:
            ds.solve_sequential(stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    reports.clear()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    barrier()
    step_ms = [a.elapsed_time(b) for a, b in ev]
    t_max = t_local
    if dist is not None:
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    out = ds.results()
    status, iters = out['status'], out['iters']
    solved_mask = status != 1
    solved = int(solved_mask.sum())
    total_solved = solved
    if dist is not None:
        ts = torch.tensor([solved], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_solved = int(ts.item())
    value = total_solved * args.steps / t_max
    ms_per_step = 1e3 * t_max / args.steps

    name, lds, thr = capi.last_launch_info()
    rep = reports[-1] if reports else None
    how = (f'chunked: {rep["n_chunks"]} concurrent chunks (1 workgroup each), {rep["warmup"]}-frame warm-up overlap, hand-offs '
           f'verified to {rep["verify_tol"]:g} and repaired exactly' if rep else 'exact sequential chain (1 chain = 1 workgroup)')
    result = {
        'metric': 'solved mocap frames/sec (Stage-II)', 'value': round(value, 2), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'BASELINE config[1]: {F}-frame SMPL-H sequence, {M} body markers, fixed betas; {how}',
                   'mode': args.mode, 'frames_per_gpu': F, 'markers': M, 'free_vars_step1': 3 + len(solver.ids['step1']),
                   'free_vars_step2': 3 + len(solver.ids['step2']), 'sequences_per_gpu': 1},
    }
    if rep:
        result['chunking'] = dict(rep, repaired_per_step=float(np.mean([r['n_repaired'] for r in reports])))
    if rank == 0:
        nobs_mean = float(job['vis'].sum(1).mean())
        fl = solver_flops(sm.K, 3 * M, len(solver.ids['step1']), len(solver.ids['step2']), sm.NP,
                          len(solver.ids['body']), nobs_mean, iters[solved_mask, 0].sum(), iters[solved_mask, 1].sum())
        kt = float(np.mean(step_ms)) * 1e-3
        ach = fl / kt / 1e12
        result['roofline'] = {
            'kernel': name, 'bound': 'valu_f64',
            'bound_note': 'neither hbm nor mfma: float64 vector pipe, dependency/latency-bound small dense solves; HBM traffic ~44 KB/frame (PMC)',
            'achieved': round(ach, 5), 'peak': F64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / F64_VALU_PEAK_TFLOPS, 6),
            # PMC (separate FETCH_SIZE / WRITE_SIZE passes on a 120-frame chain, profiles/r01_chain_pmc.txt): 13.3 KB fetched
            # (x2 wide-load correction of the guide -> 26.5 KB) + 17.7 KB written per solved frame, incl. the one-time read of
            # the 1.76 MB attachment slice and the kernel's scratch write-backs; scaled to the frames the timed mode solves
            'traffic': int(44.2e3 * (F + (rep['n_chunks'] * rep['warmup'] if rep else 0))),
            'traffic_source': 'rocprofv3 PMC per solved frame (profiles/r01_chain_pmc.txt) x frames solved in pass 1; not collected live',
            'step_ms_hip_events': round(kt * 1e3, 3),
            'algorithmic_gflop_per_step': round(fl / 1e9, 3),
            'note': 'algorithmic = the sequential chain\'s work on the recorded frames; warm-up and repair work is overhead',
            'dogleg_iters_per_frame': round(float(iters[solved_mask, 0].sum()) / max(solved, 1), 3),
            'residual_evals_per_frame': round(float(iters[solved_mask, 1].sum()) / max(solved, 1), 3), 'lds_bytes': lds,
        }
        # ---- the literal frame order on one workgroup, and the timed mode's deviation from it over all frames
        extras = world == 1     # the reference / spread / CPU / Stage-I legs run at N = 1 only: at N > 1 the other ranks would idle at the barrier
        if extras and args.mode == 'chunked' and not args.no_sequential:
            dsq = workload.DeviceSequence(job, solver, dev)
            stream = torch.cuda.current_stream().cuda_stream
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            dsq.solve_sequential(stream)
            torch.cuda.synchronize()
            tsq = time.perf_counter() - ts0
            sq = dsq.results()
            dp = np.abs(out['fullpose'] - sq['fullpose'])[solved_mask].max(1)
            dm = out['markers_sim'][solved_mask] - sq['markers_sim'][solved_mask]
            result['sequential_chain'] = {
                'frames_per_s': round(solved / tsq, 1), 'ms': round(tsq * 1e3, 1), 'kernel': capi.last_launch_info()[0],
                'timed_mode_vs_sequential': {'max_abs_pose_diff_rad': float(dp.max()), 'frames_over_1e-4_rad': int((dp > 1e-4).sum()),
                                             'frames_over_1e-6_rad': int((dp > 1e-6).sum()),
                                             'marker_rmse_m': float(np.sqrt((dm ** 2).sum(-1).mean())),
                                             'status_identical': bool((status == sq['status']).all())}}
            result['speedup_vs_sequential_chain'] = round(value / max(solved / tsq, 1e-9) / world, 2)
            del dsq
        # ---- the same kernel with the chip full: 32 copies of the sequence in one call (BASELINE config[2] shape: many
        # sequences per GPU).  8 chunks per sequence, so warm-up is 7 % of the work and every CU carries a chain.
        if extras and args.many > 0:
            try:
                copies = [workload.DeviceSequence(job, solver, dev) for _ in range(args.many)]
                stream = torch.cuda.current_stream().cuda_stream
                workload.solve_many_chunked(copies, stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                mrep = workload.solve_many_chunked(copies, stream)
                e1.record()
                torch.cuda.synchronize()
                mt = e0.elapsed_time(e1) * 1e-3
                mout = copies[-1].results()
                dpm = np.abs(mout['fullpose'] - out['fullpose'])[solved_mask].max()
                mach = fl * args.many / mt / 1e12
                result['many_sequences'] = {
                    'sequences': args.many, 'frames': args.many * F, 'frames_per_s': round(args.many * solved / mt, 1), 'ms': round(mt * 1e3, 2),
                    'chunking': mrep, 'max_abs_pose_diff_vs_single_run_rad': float(dpm),
                    'roofline': {'kernel': name, 'bound': 'valu_f64', 'achieved': round(mach, 4), 'peak': F64_VALU_PEAK_TFLOPS,
                                 'unit': 'TFLOP/s', 'frac': round(mach / F64_VALU_PEAK_TFLOPS, 5)}}
                del copies
            except Exception as e:
                result['many_sequences'] = {'error': repr(e)}
        # ---- the same workload from other seeds: the chunk-repair pattern depends on the motion (how long the regions are in
        # which a fresh start sits in another basin), so the headline seed is not the whole story
        if extras and args.mode == 'chunked' and args.spread_seeds:
            spread = {}
            try:
                for sd in [int(x) for x in args.spread_seeds.split(',') if x.strip()]:
                    job2 = workload.make_job('smplh', n_frames=args.frames, n_markers=args.markers, seed=sd)
                    solver2 = workload.make_solver(job2)
                    ds2 = workload.DeviceSequence(job2, solver2, dev)
                    stream2 = torch.cuda.current_stream().cuda_stream
                    ds2.solve_chunked(stream2, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    rep2 = ds2.solve_chunked(stream2, num_chunks=args.chunks, warmup=args.chunk_warmup, verify_tol=args.verify_tol)
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t2
                    n2 = int((ds2.results()['status'] != 1).sum())
                    spread[str(sd)] = {'frames_per_s': round(n2 / dt2, 1), 'ms_per_step': round(dt2 * 1e3, 2),
                                       'n_repaired': rep2['n_repaired'], 'repair_rounds': rep2['repair_rounds']}
                    del ds2, solver2, job2
                result['other_seeds'] = spread
            except Exception as e:
                result['other_seeds'] = {'error': repr(e)}
        # ---- full-mesh LBS export kernel (the kernel the HBM-roofline target names)
        try:
            import ctypes as C
            Fl = min(args.lbs_frames, F)
            pose32 = ds.pose[:Fl].to(torch.float32).contiguous()
            trans32 = ds.trans[:Fl].to(torch.float32).contiguous()
            verts = torch.empty((Fl, sm.V, 3), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                solver.dev.lbs_forward_device(Fl, pose32.data_ptr(), trans32.data_ptr(), verts.data_ptr(), C.c_void_p(stream))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                solver.dev.lbs_forward_device(Fl, pose32.data_ptr(), trans32.data_ptr(), verts.data_ptr(), C.c_void_p(stream))
            e1.record()
            torch.cuda.synchronize()
            lt = e0.elapsed_time(e1) * 1e-3 / reps
            Kj = sm.K
            # algorithmic bytes: 12 V out + pose/trans in per frame, plus ONE read of the model in the precision the kernel
            # consumes it (f16 posedirs, f32 rest vertices, sparse skinning weights as (joint, weight) pairs)
            model_bytes = 2 * 3 * sm.V * 9 * (Kj - 1) + 12 * sm.V + 8 * 4 * sm.V
            bytes_alg = Fl * (12 * sm.V + 4 * sm.NP + 12) + model_bytes
            result['roofline_lbs'] = {'kernel': 'k_lbs_mfma (+ k_lbs_prep)', 'bound': 'hbm', 'dtype': 'f32 out; f16-operand / f32-accumulate MFMA correctives', 'achieved': round(bytes_alg / lt / 1e9, 1),
                                      'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(bytes_alg / lt / 1e9 / HBM_PEAK_GBS, 4),
                                      # PMC at F=4000 (profiles/r01_lbs_pmc.txt): FETCH_SIZE 234.8 MB (x2 -> 469.6 MB), WRITE_SIZE 430.7 MB
                                      # (+ k_lbs_prep 2 x 1.2 + 13.4 MB)
                                      'traffic': int((2 * 234.794e6 + 430.65e6 + 2 * 1.175e6 + 13.376e6) * Fl / 4000.0),
                                      'traffic_source': 'rocprofv3 PMC at F=4000 (profiles/r01_lbs_pmc.txt), scaled by frames; not collected live',
                                      'frames': Fl, 'kernel_ms': round(lt * 1e3, 3),
                                      'frames_per_s': round(Fl / lt, 1)}
        except Exception as e:   # the LBS leg must never take the headline number down
            result['roofline_lbs'] = {'error': repr(e)}
        # ---- CPU baseline: the NumPy oracle ("port") on a bounded sample of the same workload; parity on that sample
        if extras and not args.no_cpu:
            from oracle import stageii_oracle as so
            S = min(args.cpu_sample, F)
            seq = job['seq']
            m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs,
                                      weights=sm.weights, J_regressor=sm.J_regressor, parents=sm.parents,
                                      body_dof=sm.body_dof, hand_dof=sm.hand_dof, hands_mean=sm.hands_mean,
                                      selected_components=sm.selected_components), job['betas'])
            pr = so.prepare_gmm_prior(seq['gmm'], 63)
            can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
            closest, coef = so.transformed_coeffs(can, job['markers_latent'])
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
            n_ref = len(ref['frame_ids'])
            result['cpu_baseline'] = {'value': round(n_ref / tc, 2), 'unit': 'frames/s', 'cores': int(blas_threads),
                                      'kind': 'port', 'host_cores_visible': os.cpu_count(),
                                      'sample': f'first {S} frames of the same sequence, NumPy float64 oracle '
                                                f'(lean marker-subset mode), single process, BLAS pinned to '
                                                f'{blas_threads} thread(s), {tc:.1f} s'}
            gp = out['fullpose'][:S][status[:S] != 1]
            gm = out['markers_sim'][:S]
            sqd = []
            for i, t in enumerate(ref['frame_ids']):
                sqd.append(((gm[t][job['vis'][t]] - ref['markers_sim'][i]) ** 2).sum(1))
            dpo = np.abs(gp - ref['fullpose']).max(1)
            result['parity'] = {'against': 'oracle sequential chain', 'frames': int(n_ref),
                                'max_abs_pose_diff_rad': float(dpo.max()), 'frames_over_1e-4_rad': int((dpo > 1e-4).sum()),
                                'marker_rmse_m': float(np.sqrt(np.concatenate(sqd).mean())),
                                'tolerance': {'pose_rad': 1e-4, 'marker_rmse_m': 1e-3}}
            result['speedup_vs_cpu_port'] = round(value / max(n_ref / tc, 1e-9), 1)
        # ---- Stage-I leg (BASELINE config 4's calibration part): 12 picked frames, 53 markers, 10 betas on a
        #      triangulated SMPL-H-sized body; the joint solve on the GPU beside the NumPy oracle on the host
        if extras and not args.no_stagei:
            try:
                from moshpp_amd import capi
                pb1, dev1, pr1, kw1 = workload.make_stagei_job()
                capi.stagei_solve_host(dev1, pr1, **kw1)
                ts = []
                for _ in range(3):
                    t1 = time.perf_counter(); o1 = capi.stagei_solve_host(dev1, pr1, **kw1); ts.append(time.perf_counter() - t1)
                leg = {'workload': f"12 frames, 53 markers, 10 betas, V={pb1['model']['v_template'].shape[0]}, {len(pb1['faces'])} triangles",
                       'unknowns': int(3 * 12 + 3 * 53 + 12 * len(kw1['pose_ids']) + 10), 'seconds': round(float(np.median(ts)), 4),
                       'dogleg_iterations': o1['iters']}
                try:    # the arrow-structured solver (per-frame elimination + Schur complement), still opt-in: MOSHII_S1_SOLVER=schur
                    os.environ['MOSHII_S1_SOLVER'] = 'schur'
                    capi.stagei_solve_host(dev1, pr1, **kw1)
                    t1 = time.perf_counter(); o2 = capi.stagei_solve_host(dev1, pr1, **kw1); leg['seconds_schur_solver'] = round(time.perf_counter() - t1, 4)
                    leg['schur_vs_dense_max_abs_betas_diff'] = float(np.abs(o2['betas'] - o1['betas']).max())
                except Exception as e:
                    leg['seconds_schur_solver'] = repr(e)
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
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
