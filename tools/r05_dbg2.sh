#!/bin/bash
export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "ill_conditioned or competing" 2>&1 | grep -v "amdgpu.ids" | tail -30 | cut -c1-700
