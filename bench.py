#!/usr/bin/env python
"""Stage-II throughput bench (BASELINE.json metric: solved mocap frames/sec).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full Stage-II pass (moshii_chain_solve) over one synthetic sequence of BASELINE config[1]
(4000-frame SMPL-H, 53 body markers, fixed betas, exact sequential-chain semantics of the reference), with
observations already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank solves its own sequence of the same shape -- the path has no data-path collective -- and the
job-level frames/s is reported ("scaling": "weak").

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel, k_chain_solve), `roofline_lbs`
(full-mesh LBS export kernel), `cpu_baseline` (the NumPy oracle on a bounded sample, host cores of this box),
`parity` (GPU vs oracle on that sample).
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

F64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector/matrix FP64 (AMD spec sheet; half the 157.3 TF FP32 rate of MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def solver_flops(K, Nv, n1, n2, NP, npose, nobs_mean, iters, fevals, first_frame_rounds=0):
    """Algorithmic FLOPs of the solve per SURVEY.md 8(d) / DESIGN.md:
    forward 2*3Nv*9(K-1) + 2*Nv*K*12 per residual evaluation; per Jacobian 3Nv*n*30 + normal equations
    2*m*n^2/2 + Cholesky n^3/3, with m = 3*nobs + (npose+1) + NP rows."""
    fwd = 2.0 * 3 * Nv * 9 * (K - 1) + 2.0 * Nv * K * 12
    m = 3.0 * nobs_mean + (npose + 1 if npose else 0) + NP
    n = 3 + 0.5 * (n1 + n2)
    per_iter = 3.0 * Nv * n * 30 + m * n * n + n ** 3 / 3.0
    return float(fevals) * fwd + float(iters) * per_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=4000)
    ap.add_argument('--markers', type=int, default=53)
    ap.add_argument('--cpu-sample', type=int, default=200, help='frames of the workload timed on the CPU oracle')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--lbs-frames', type=int, default=2000)
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
    capi.load()
    check = capi.check
    lib = capi.load()
    check(lib.moshii_set_device(local_rank))

    # ---- workload: BASELINE config[1]; each rank its own seeded sequence of the same shape
    job = workload.make_job('smplh', n_frames=args.frames, n_markers=args.markers, seed=1000 + rank)
    solver = workload.make_solver(job)
    sm = job['sm']
    F, M = job['vis'].shape
    dev = torch.device('cuda', local_rank)
    obs_d = torch.from_numpy(np.ascontiguousarray(job['obs'])).to(dev)
    vis_d = torch.from_numpy(np.ascontiguousarray(job['vis'].astype(np.uint8))).to(dev)
    out_pose = torch.zeros((F, sm.NP), dtype=torch.float64, device=dev)
    out_full = torch.zeros((F, 3 * sm.K), dtype=torch.float64, device=dev)
    out_trans = torch.zeros((F, 3), dtype=torch.float64, device=dev)
    out_msim = torch.zeros((F, M, 3), dtype=torch.float64, device=dev)
    out_errs = torch.zeros((F, 4), dtype=torch.float64, device=dev)
    out_iters = torch.zeros((F, 2), dtype=torch.int32, device=dev)
    out_status = torch.zeros((F,), dtype=torch.int32, device=dev)
    desc = (capi.ChainDesc * 1)()
    d = desc[0]
    d.attach = solver.attach.handle
    d.F = F
    d.first_frame_schedule = 1
    d.obs = obs_d.data_ptr(); d.vis = vis_d.data_ptr()
    d.pose = out_pose.data_ptr(); d.fullpose = out_full.data_ptr(); d.trans = out_trans.data_ptr()
    d.markers_sim = out_msim.data_ptr(); d.errs = out_errs.data_ptr(); d.iters = out_iters.data_ptr()
    d.status = out_status.data_ptr()
    import ctypes as C
    opts = solver.opts[0]
    prior_h = solver.prior.handle if solver.prior is not None else None

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.moshii_chain_solve(solver.dev.handle, prior_h, C.byref(opts), 1, desc, capi.BUFFERS_DEVICE,
                                     C.c_void_p(stream)))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
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
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    t_max = t_local
    if dist is not None:
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    status = out_status.cpu().numpy()
    iters = out_iters.cpu().numpy()
    solved = int((status != 1).sum())
    total_solved = solved
    if dist is not None:
        ts = torch.tensor([solved], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_solved = int(ts.item())
    value = total_solved * args.steps / t_max
    ms_per_step = 1e3 * t_max / args.steps

    result = {
        'metric': 'solved mocap frames/sec (Stage-II)', 'value': round(value, 2), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'BASELINE config[1]: {F}-frame SMPL-H sequence, {M} body markers, fixed betas, '
                               f'exact sequential chain (1 chain = 1 workgroup per GPU)',
                   'frames_per_gpu': F, 'markers': M, 'free_vars_step1': 3 + len(solver.ids['step1']),
                   'free_vars_step2': 3 + len(solver.ids['step2']), 'sequences_per_gpu': 1},
    }
    if rank == 0:
        name, lds, thr = capi.last_launch_info()
        nobs_mean = float(job['vis'].sum(1).mean())
        fl = solver_flops(sm.K, 3 * M, len(solver.ids['step1']), len(solver.ids['step2']), sm.NP,
                          len(solver.ids['body']), nobs_mean, iters[:, 0].sum(), iters[:, 1].sum())
        kt = float(np.mean(kern_ms)) * 1e-3
        ach = fl / kt / 1e12
        result['roofline'] = {
            'kernel': name, 'bound': 'valu_f64 (dependency/latency-bound: ONE workgroup on 1 of 256 CUs)',
            'achieved': round(ach, 5), 'peak': F64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / F64_VALU_PEAK_TFLOPS, 6),
            'frac_of_one_cu': round(ach / (F64_VALU_PEAK_TFLOPS / 256.0), 4),
            'traffic': None, 'kernel_ms': round(kt * 1e3, 3), 'algorithmic_gflop_per_launch': round(fl / 1e9, 3),
            'dogleg_iters_per_frame': round(float(iters[:, 0].sum()) / max(solved, 1), 3),
            'residual_evals_per_frame': round(float(iters[:, 1].sum()) / max(solved, 1), 3), 'lds_bytes': lds,
        }
        # ---- full-mesh LBS export kernel (the kernel the HBM-roofline target names)
        try:
            Fl = args.lbs_frames
            pose32 = out_pose[:Fl].to(torch.float32).contiguous()
            trans32 = out_trans[:Fl].to(torch.float32).contiguous()
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
            model_bytes = 12 * sm.V * (1 + 9 * (Kj - 1)) + 4 * sm.V * Kj
            bytes_alg = Fl * (12 * sm.V + 4 * 3 * Kj + 12) + model_bytes
            result['roofline_lbs'] = {'kernel': 'lbs_forward_f32', 'bound': 'hbm', 'achieved': round(bytes_alg / lt / 1e9, 1),
                                      'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(bytes_alg / lt / 1e9 / HBM_PEAK_GBS, 4),
                                      'traffic': None, 'frames': Fl, 'kernel_ms': round(lt * 1e3, 3),
                                      'frames_per_s': round(Fl / lt, 1)}
        except Exception as e:   # the LBS leg must never take the headline number down
            result['roofline_lbs'] = {'error': repr(e)}
        # ---- CPU baseline: the NumPy oracle ("port") on a bounded sample of the same workload
        if not args.no_cpu:
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
            tc0 = time.perf_counter()
            ref = so.stageii_chain(m, pr, closest, coef, job['obs'][:S], job['vis'][:S], 'smplh')
            tc = time.perf_counter() - tc0
            n_ref = len(ref['frame_ids'])
            try:
                import threadpoolctl
                blas_threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
            except Exception:
                blas_threads = os.cpu_count()
            result['cpu_baseline'] = {'value': round(n_ref / tc, 2), 'unit': 'frames/s', 'cores': int(blas_threads),
                                      'kind': 'port', 'host_cores_visible': os.cpu_count(),
                                      'sample': f'first {S} frames of the same sequence, NumPy float64 oracle '
                                                f'(lean marker-subset mode), single process, {tc:.1f} s'}
            gp = out_full[:S].cpu().numpy()[status[:S] != 1]
            gm = out_msim[:S].cpu().numpy()
            sq = []
            for i, t in enumerate(ref['frame_ids']):
                sq.append(((gm[t][job['vis'][t]] - ref['markers_sim'][i]) ** 2).sum(1))
            result['parity'] = {'frames': int(n_ref), 'max_abs_pose_diff_rad': float(np.abs(gp - ref['fullpose']).max()),
                                'marker_rmse_m': float(np.sqrt(np.concatenate(sq).mean())),
                                'tolerance': {'pose_rad': 1e-4, 'marker_rmse_m': 1e-3}}
            result['speedup_vs_cpu_port'] = round(value / max(n_ref / tc, 1e-9), 1)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
