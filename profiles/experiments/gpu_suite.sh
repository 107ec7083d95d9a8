#!/bin/bash
# the whole GPU suite in one gpurun call (summary + failures)
cd /root/repo
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu -rf > gpurun_out/final/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|Fatal|fault" gpurun_out/final/pytest_gpu_full.log | tail -12 | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
