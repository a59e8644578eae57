#!/bin/bash
export PYTHONPATH=.
for body in mesh shuffled; do for F in 701 1357; do LBS_BODY=$body python tools/r05_lbs_soak_dbg.py $F 2>&1 | grep -v amdgpu.ids | cut -c1-200 | grep rep; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lbs" 2>&1 | tail -4
for body in mesh shuffled; do LBS_BODY=$body python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids; done
