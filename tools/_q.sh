cd /root/repo; export PYTHONPATH=/root/repo
O=gpurun_out/q1; mkdir -p $O
( MOSHII_COOP=1 python tools/chain_time.py 400 | tail -1; python tools/chain_time.py 400 | tail -1;
  MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 timeout 200 python tools/prof_chain.py 400 smplh | grep -i "back-sub\|chol\|wall" ) > $O/out.txt 2>&1
