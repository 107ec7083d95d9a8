mkdir -p gpurun_out/r2c15
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r2c15/pytest_full.log
cat gpurun_out/r2c15/pytest_full.log
timeout 600 python bench.py --steps 64 > gpurun_out/r2c15/bench.json 2> gpurun_out/r2c15/bench.err; tail -c 3000 gpurun_out/r2c15/bench.json; tail -3 gpurun_out/r2c15/bench.err
