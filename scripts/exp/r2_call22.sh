#!/bin/bash
# call 22: wave-specialised prefill GEMM A/B + parity
mkdir -p gpurun_out/r2c22
cd /root/repo
{
echo "== variant 0 (ping-pong), T=512"; MRS_GEMM_VARIANT=0 timeout 300 python scripts/bench_gemm.py --big --t 512 --types q4_k
echo "== variant 1 (wave-specialised), T=512"; MRS_GEMM_VARIANT=1 timeout 300 python scripts/bench_gemm.py --big --t 512 --types q4_k,q6_k
echo "== variant 1, T=2048"; MRS_GEMM_VARIANT=1 timeout 300 python scripts/bench_gemm.py --big --t 2048 --types q4_k
echo "== variant 0, T=2048"; MRS_GEMM_VARIANT=0 timeout 300 python scripts/bench_gemm.py --big --t 2048 --types q4_k
} > gpurun_out/r2c22/gemm.log 2>&1
cat gpurun_out/r2c22/gemm.log
timeout 600 python -m pytest tests/test_gemm.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 64 2>&1 | tail -1 | cut -c1-1500
