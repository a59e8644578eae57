#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
for rep in 1 2; do for v in head new; do
  if [ $v = new ]; then L=""; else L="MOSHII_LIB=$PWD/moshpp_amd/ab/libmoshii_$v.so"; fi
  env $L python bench.py --no-cpu --no-stagei --no-strong --no-sequential --steps 2 --warmup 1 --seeds 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config3']; print('$v', 'config3', c['frames_per_s'], c['us_per_frame_per_chain'], c['kernel'], 'rmse', c['marker_rmse_m'])"
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "free_shape or config3 or expression or extended" 2>&1 | tail -3
