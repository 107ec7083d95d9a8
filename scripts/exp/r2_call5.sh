mkdir -p gpurun_out/r2c5
timeout 900 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -s 2>&1 | tail -40 > gpurun_out/r2c5/pytest_model.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_dec_model.py 2>&1 | tail -15 > gpurun_out/r2c5/pytest_full.log
cat gpurun_out/r2c5/pytest_model.log gpurun_out/r2c5/pytest_full.log
