mkdir -p gpurun_out/r2c7
timeout 600 python -m pytest tests/test_llama_runner.py -m gpu -q -p no:cacheprovider -x -k "mfma_prefill" > gpurun_out/r2c7/prefill.log 2>&1
tail -40 gpurun_out/r2c7/prefill.log | cut -c1-250
timeout 600 python -m pytest "tests/test_mmvq.py::test_plain_all_types_all_batches" -m gpu -q -p no:cacheprovider -x > gpurun_out/r2c7/mmvq.log 2>&1
tail -5 gpurun_out/r2c7/mmvq.log | cut -c1-250
timeout 900 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "Error|position|worst|passed|failed|FAILED" | head -30 > gpurun_out/r2c7/pytest_model.log
cat gpurun_out/r2c7/pytest_model.log
