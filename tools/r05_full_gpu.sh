#!/bin/bash
# round 5: the whole GPU test tier + the default bench line (what the driver runs at round end)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
tail -3 gpurun_out/r05_gpu_tests.txt; head -c 1500 gpurun_out/r05_bench_line.json; tail -3 gpurun_out/r05_bench.err
