#!/bin/bash
# quick device check of the export kernel: a few parity tests, timings with / without stagger, phase truncations, stamps
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lbs" 2>&1 | tail -4
for body in mesh shuffled; do LBS_BODY=$body LBS_CHECK=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids; done
echo "# stagger 0"; LBS_BODY=mesh MOSHII_LBS_STAGGER=0 timeout 300 python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids
echo "# stop=1 (prep + k-loop only), stagger 0"; LBS_BODY=mesh MOSHII_LBS_STOP=1 MOSHII_LBS_STAGGER=0 timeout 300 python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids
echo "# stop=2 (no global stores)"; LBS_BODY=mesh MOSHII_LBS_STOP=2 timeout 300 python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids
echo "# stamps"; LBS_BODY=mesh MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh 2>&1 | grep -v amdgpu.ids | cut -c1-640
