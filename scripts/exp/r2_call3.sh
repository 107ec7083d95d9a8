mkdir -p gpurun_out/r2c3
timeout 300 python scripts/bench_dec.py > gpurun_out/r2c3/bench_dec.log 2>&1
timeout 300 python -m pytest tests/test_dec_engine.py tests/test_zz_isq_kquants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r2c3/pytest.log
cat gpurun_out/r2c3/bench_dec.log gpurun_out/r2c3/pytest.log
