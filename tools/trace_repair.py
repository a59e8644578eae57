import os, sys, time
sys.path.insert(0, '.')
import torch
from moshpp_amd import workload
dev = torch.device('cuda', 0)
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
stream = torch.cuda.current_stream().cuda_stream
ds.solve_chunked(stream)
os.environ['MOSHII_TRACE_REPAIR'] = '1'
torch.cuda.synchronize(); t0 = time.perf_counter()
rep = ds.solve_chunked(stream)
torch.cuda.synchronize(); print('ms', (time.perf_counter() - t0) * 1e3, rep)
