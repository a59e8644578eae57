#!/bin/bash
export PYTHONPATH=.
echo "# stamps"; LBS_BODY=mesh MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh 2>&1 | grep -v amdgpu.ids | cut -c1-700
echo "# stamps, stagger 0"; LBS_BODY=mesh MOSHII_LBS_STOP=16 MOSHII_LBS_STAGGER=0 timeout 300 python tools/lbs_bench.py 4000 5 smplh 2>&1 | grep -v amdgpu.ids | cut -c1-700
for F in 41 128; do LBS_BODY=mesh LBS_CHECK=1 timeout 300 python tools/lbs_bench.py $F 5 smplh 2>&1 | grep -v amdgpu.ids; done
