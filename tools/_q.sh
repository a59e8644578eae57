cd /root/repo; export PYTHONPATH=/root/repo
O=gpurun_out/q6; mkdir -p $O
( MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 timeout 200 python tools/prof_config3.py 40 80 | grep -i "wall\|ldl\|back\|exchange\|other";
  timeout 600 python tools/config3_copies.py 400 | tail -3;
  timeout 300 python tools/coop_time.py 300 smplh 1 --groups=4,6,8 --fracs=0 2>&1 | tail -5;
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 ) > $O/out.txt 2>&1
