#!/bin/bash
# round-3 GPU session for the LBS export kernel: parity tests, timings per variant / truncation, kernel trace.  Output: gpurun_out/r03_lbs/
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_lbs; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k lbs > $O/pytest_lbs.txt 2>&1; tail -5 $O/pytest_lbs.txt
{
for mt in smplh smplx; do python tools/lbs_bench.py 4000 10 $mt; done
python tools/lbs_bench.py 2000 10 smplh
python tools/lbs_bench.py 50000 3 smplh
echo "# no DMA"; MOSHII_LBS_NO_DMA=1 python tools/lbs_bench.py 4000 10 smplh
echo "# stop=1 (prep + k-loop)"; MOSHII_LBS_STOP=1 python tools/lbs_bench.py 4000 10 smplh
echo "# stop=2 (no stores)"; MOSHII_LBS_STOP=2 python tools/lbs_bench.py 4000 10 smplh
echo "# plain f32 kernel"; MOSHII_LBS_PLAIN=1 python tools/lbs_bench.py 4000 3 smplh
} > $O/timings.txt 2>&1
cat $O/timings.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/trace -o lbs -- python /root/repo/tools/lbs_bench.py 4000 10 smplh > /root/repo/$O/rocprof_stdout.txt 2>&1
cd /root/repo; find $O/trace -name "*kernel_stats*" | head -3; for f in $(find $O/trace -name "*kernel_stats.csv"); do head -8 $f; done
find $O/trace -name "*.db" -delete 2>/dev/null; find $O/trace -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
