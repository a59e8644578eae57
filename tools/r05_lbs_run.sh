#!/bin/bash
# round-5 LBS export kernel: correctness on the device, then timings / stamps / stagger sweep.  Output: gpurun_out/r05_lbs_*.txt
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lbs" 2>&1 | tail -15 > gpurun_out/r05_lbs_tests.txt
{
for body in mesh shuffled; do
  LBS_BODY=$body LBS_CHECK=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh
done
for sg in 0 2 3 5 7; do echo "# stagger $sg"; LBS_BODY=mesh MOSHII_LBS_STAGGER=$sg timeout 300 python tools/lbs_bench.py 4000 20 smplh; done
echo "# stop=1 (prep + k-loop only)"; LBS_BODY=mesh MOSHII_LBS_STOP=1 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# stop=1, no stagger"; LBS_BODY=mesh MOSHII_LBS_STOP=1 MOSHII_LBS_STAGGER=0 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# stop=2 (no global stores)"; LBS_BODY=mesh MOSHII_LBS_STOP=2 timeout 300 python tools/lbs_bench.py 4000 20 smplh
echo "# stamps"; LBS_BODY=mesh MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh
echo "# stamps shuffled"; LBS_BODY=shuffled MOSHII_LBS_STOP=16 timeout 300 python tools/lbs_bench.py 4000 5 smplh
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 4000 20 smplx
LBS_BODY=mesh timeout 300 python tools/lbs_bench.py 50000 5 smplh
} > gpurun_out/r05_lbs_timings.txt 2>&1
cd /tmp && export TMPDIR=/tmp
LBS_BODY=mesh timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_lbs_prof -o lbs -- python $GRAFT_REPO_ROOT/tools/lbs_bench.py 4000 20 smplh > $GRAFT_REPO_ROOT/gpurun_out/r05_lbs_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05_lbs_prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r05_lbs_kernel_stats.csv
tail -5 gpurun_out/r05_lbs_tests.txt; cat gpurun_out/r05_lbs_timings.txt | cut -c1-600
