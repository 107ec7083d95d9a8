mkdir -p gpurun_out/r2c2
timeout 600 python -m pytest tests/test_dec_engine.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r2c2/pytest_dec.log
timeout 300 python scripts/bench_dec.py --old > gpurun_out/r2c2/bench_dec.log 2>&1
timeout 120 python scripts/exp/dbg_isq_legacy.py > gpurun_out/r2c2/dbg_isq.log 2>&1
cat gpurun_out/r2c2/pytest_dec.log gpurun_out/r2c2/bench_dec.log gpurun_out/r2c2/dbg_isq.log
