#!/bin/bash
O=gpurun_out/r3c8; mkdir -p $O
cd /root/repo
(timeout 1500 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -q -m gpu 2>&1 | tail -25) | tee $O/pytest.log
(timeout 900 python bench.py 2>&1 | tail -1) | tee $O/bench_default.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) | tee $O/smoke.log
