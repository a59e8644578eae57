#!/bin/bash
# Round-2 artefact collection on the GPU box (one gpurun call): tests, smoke, bench, rocprofv3 stats, PMC passes, phase breakdown.
# Output under gpurun_out/r02f/; the summaries are copied into profiles/ by hand.
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest_gpu.txt
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
MOSHII_LIB=moshpp_amd/libmoshii_prof.so timeout 200 python tools/prof_chain.py 400 smplh > $O/phase_breakdown.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu --no-stagei > $O/bench_line_under_rocprof.json 2> $O/rocprof_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu --no-stagei --no-strong --no-sequential --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> $O/pmc_${c}_err.txt
  python tools/pmc_summary.py $O/pmc_$c k_chain_solve > $O/pmc_$c.txt 2>&1
done
ls $O
