#!/bin/bash
# round 5: SQ-side counters of the sequential chain (cooperative, 6 workgroups; and one workgroup: MOSHII_COOP=1), 400 frames
# (separate rocprofv3 --pmc passes with --kernel-trace only) -> gpurun_out/r05/chain_sq_counters.txt
cd /root/repo; export TMPDIR=/tmp PYTHONPATH=/root/repo
O=gpurun_out/r05; mkdir -p $O
{
echo "# k_chain_solve, SMPL-H / 53 markers, one sequential chain of 400 frames (tools/chain_time.py; 3 timed + 1 short launch per pass): rocprofv3 --kernel-trace --pmc <set>, sums over the launches of a pass"
for coop in auto 1; do
echo "## MOSHII_COOP=$coop"
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && MOSHII_COOP=$coop timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/cq$i -- python /root/repo/tools/chain_time.py 400 smplh > /root/repo/$O/cq$i.log 2>&1)
  python tools/pmc_summary.py $O/cq$i k_chain 2>&1 | sed 's/^void moshii:://' | cut -c1-140
  rm -rf $O/cq$i $O/cq$i.log
done
done
} > $O/chain_sq_counters.txt
cat $O/chain_sq_counters.txt
