"""Dump the GPU sequential and chunk-parallel fullpose of one seeded bench-shaped sequence: python tools/dump_seq.py <seed> <out.npz>"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload
seed, out = int(sys.argv[1]), sys.argv[2]
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=seed)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
ds.solve_sequential(stream); torch.cuda.synchronize()
seq = ds.results()
ds.solve_chunked(stream, verify_tol=1e-11); torch.cuda.synchronize()
chk = ds.results()
np.savez_compressed(out, seq=seq['fullpose'], chk=chk['fullpose'], seq_iters=seq['iters'], chk_iters=chk['iters'])
