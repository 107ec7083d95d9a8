mkdir -p gpurun_out/r2c8
timeout 900 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "Error|position|worst|passed|failed|FAILED" | head -30 > gpurun_out/r2c8/pytest_model.log
cat gpurun_out/r2c8/pytest_model.log
timeout 900 python -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider --deselect tests/test_dec_model.py > gpurun_out/r2c8/pytest_full.log 2>&1
grep -n "Fatal\|FAILED" gpurun_out/r2c8/pytest_full.log | head; tail -3 gpurun_out/r2c8/pytest_full.log | cut -c1-300
