#!/bin/bash
O=gpurun_out/r2c29; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gemm.py tests/test_zz_moe_prefill.py tests/test_hqq.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do
  echo "== MRS_MOE_PREFILL_MFMA=$v"
  MRS_MOE_PREFILL_MFMA=$v timeout 900 python scripts/bench_mixtral.py --steps 64 2>&1 | tail -1 | cut -c1-700
done 2>&1 | tee $O/mixtral.log
echo "== 2048-token prompt, MFMA route"
timeout 900 python scripts/bench_mixtral.py --steps 32 --prompt-len 2048 2>&1 | tail -1 | cut -c1-700 | tee -a $O/mixtral.log
