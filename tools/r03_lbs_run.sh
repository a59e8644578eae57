#!/bin/bash
# round-3 GPU session for the LBS export kernel: parity tests, timings per variant / truncation, kernel trace, PMC passes.
# Output: gpurun_out/r03_lbs/   (usage: tools/r03_lbs_run.sh [nopmc])
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_lbs; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k lbs > $O/pytest_lbs.txt 2>&1; tail -5 $O/pytest_lbs.txt
{
for mt in smplh smplx; do python tools/lbs_bench.py 4000 10 $mt; done
python tools/lbs_bench.py 2000 10 smplh
python tools/lbs_bench.py 50000 3 smplh
echo "# stop=1 (prep + k-loop)"; MOSHII_LBS_STOP=1 python tools/lbs_bench.py 4000 10 smplh
echo "# stop=2 (no stores)"; MOSHII_LBS_STOP=2 python tools/lbs_bench.py 4000 10 smplh
echo "# stamps (clock64 ticks of workgroup 0, wave 0)"; MOSHII_LBS_STOP=16 python tools/lbs_bench.py 4000 2 smplh
echo "# stamps, four waves"; MOSHII_LBS_WAVES=4 MOSHII_LBS_STOP=16 python tools/lbs_bench.py 4000 2 smplh
} > $O/timings.txt 2>&1
grep -v amdgpu.ids $O/timings.txt
export PYTHONPATH=/root/repo
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace -o lbs -- python /root/repo/tools/lbs_bench.py 4000 10 smplh > /root/repo/$O/rocprof_stdout.txt 2>&1)
find $O/trace -type f | head; for f in $(find $O/trace -name "*stats*.csv"); do cp $f $O/; head -8 $f; done
rm -rf $O/trace
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|SQC_" | head -40 > $O/counters_icache.txt; cat $O/counters_icache.txt
[ "$1" = "nopmc" ] && exit 0
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE SQ_WAVES SQ_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/p$i -- python /root/repo/tools/lbs_bench.py 4000 3 > /root/repo/$O/p$i.log 2>&1)
  python tools/pmc_summary.py $O/p$i k_lbs > $O/pmc_$i.txt 2>&1
  cat $O/pmc_$i.txt
  rm -rf $O/p$i
done
