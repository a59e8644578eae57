cd /root/repo; export PYTHONPATH=/root/repo
O=gpurun_out/c3p; mkdir -p $O
( echo "# MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 python tools/prof_config3.py 40 80";
  MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=1 timeout 200 python tools/prof_config3.py 40 80;
  echo; echo "# MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 python tools/prof_config3.py 40 80  (rank 0 of the group)";
  MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=8 timeout 200 python tools/prof_config3.py 40 80 ) > $O/config3_phases.txt 2>&1
