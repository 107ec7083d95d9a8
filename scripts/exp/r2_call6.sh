mkdir -p gpurun_out/r2c6
timeout 900 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -s -k "north_star" 2>&1 | grep -E "AssertionError|position|worst|passed|failed" | head -20 > gpurun_out/r2c6/pytest_model.log
timeout 900 python -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider --deselect tests/test_dec_model.py > gpurun_out/r2c6/pytest_full.log 2>&1
grep -n "Fatal\|Segmentation\|Aborted\|core" gpurun_out/r2c6/pytest_full.log | head; grep -n "PASSED\|FAILED" gpurun_out/r2c6/pytest_full.log | tail -3; grep -A25 "Fatal Python" gpurun_out/r2c6/pytest_full.log | head -60
cat gpurun_out/r2c6/pytest_model.log
