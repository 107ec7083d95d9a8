mkdir -p gpurun_out/r2c1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r2c1/pytest_full.log
bash scripts/exp/occ4_sweep.sh > gpurun_out/r2c1/occ4.log 2>&1
(timeout 200 python scripts/bench_gemm.py --mmq --types q4_k; timeout 200 python scripts/bench_gemm.py --big --types q4_k) > gpurun_out/r2c1/gemm.log 2>&1
(timeout 300 python scripts/bench_mixtral.py --prompt-path grouped --prompt-len 512; timeout 300 python scripts/bench_mixtral.py --prompt-path chunked --prompt-len 512) > gpurun_out/r2c1/mixtral.log 2>&1
tail -30 gpurun_out/r2c1/pytest_full.log; cat gpurun_out/r2c1/occ4.log
