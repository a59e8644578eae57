"""Several sequences in ONE moshii_sequence_solve call (chunks of different sequences side by side) against each sequence's own
sequential chain, repeated: python tools/many_seq_check.py [repeats]."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
seeds = [1000, 71, 5, 123, 7, 2024]
job0 = workload.make_job('smplh', 1500, 53, seed=seeds[0])
solver = workload.make_solver(job0)
jobs = [job0] + [workload.make_job('smplh', 1500 - 100 * i, 53, seed=sd) for i, sd in enumerate(seeds[1:], 1)]
seqs = [workload.DeviceSequence(j, solver, dev) for j in jobs]
refs = []
for s in seqs:
    s.solve_sequential(stream); torch.cuda.synchronize(); refs.append(s.results()['fullpose'].copy())
for r in range(reps):
    rep = workload.solve_many_chunked(seqs, stream, verify_tol=1e-9)
    torch.cuda.synchronize()
    worst = max(float(np.abs(s.results()['fullpose'] - ref).max()) for s, ref in zip(seqs, refs))
    print(f'run {r}: {len(seqs)} sequences, worst |chunked - sequential| {worst:.2e} rad, report {rep}', flush=True)
    assert worst < 1e-7
