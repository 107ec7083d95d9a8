#!/bin/bash
O=gpurun_out/r2c28; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_hqq.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/bench_hqq.py 2>&1 | grep -v amdgpu.ids | tee $O/hqq.log
