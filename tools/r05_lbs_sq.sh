#!/bin/bash
# round 5: SQ-side counters of the LBS export (separate rocprofv3 --pmc passes with --kernel-trace only) -> gpurun_out/r05/lbs_sq_counters.txt
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r05; mkdir -p $O
{
echo "# k_lbs_export / k_lbs_prep, SMPL-H mesh-ordered body, F = 4000: rocprofv3 --kernel-trace --pmc <set> (three separate passes), per-launch means (tools/pmc_summary.py)"
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && LBS_BODY=mesh timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/sq$i -- python /root/repo/tools/lbs_bench.py 4000 3 smplh > /root/repo/$O/sq$i.log 2>&1)
  python tools/pmc_summary.py $O/sq$i k_lbs 2>&1
  rm -rf $O/sq$i $O/sq$i.log
done
} > $O/lbs_sq_counters.txt
cat $O/lbs_sq_counters.txt
