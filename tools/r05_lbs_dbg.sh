#!/bin/bash
export PYTHONPATH=.
echo "# nt stores"; LBS_BODY=mesh python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids
echo "# plain stores"; LBS_BODY=mesh MOSHII_LBS_STOP=64 python tools/lbs_bench.py 4000 20 smplh 2>&1 | grep -v amdgpu.ids
echo "# plain stores F=50000"; LBS_BODY=mesh MOSHII_LBS_STOP=64 python tools/lbs_bench.py 50000 5 smplh 2>&1 | grep -v amdgpu.ids
