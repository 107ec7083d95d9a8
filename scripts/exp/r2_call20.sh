mkdir -p gpurun_out/r2c20
(timeout 400 python bench.py --quant q8_0_isq --no-cpu-baseline --steps 128 2>&1 | tail -1) > gpurun_out/r2c20/q8_0.json
(timeout 400 python scripts/bench_mixtral.py --prompt-path grouped --prompt-len 512 2>&1 | tail -1) > gpurun_out/r2c20/mixtral.json
(MRS_DEC_PERSIST=0 timeout 400 python bench.py --no-cpu-baseline --steps 128 --prompt-len 2048 2>&1 | tail -1) > gpurun_out/r2c20/p2048.json
for f in q8_0 mixtral p2048; do echo "== $f"; cut -c1-900 gpurun_out/r2c20/$f.json; done
