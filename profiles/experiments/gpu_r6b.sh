#!/bin/bash
# round 6, call B: Q8_0 / MoE / TP exact prompt path, chained decode step, two-process p2p test on the device; default bench line; probes (XCD hand-off edges, f32 MFMA
# chain order, library bf16 GEMM at the prompt shapes); in-graph kernel trace of the chained step
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
python -c "import bench; print(bench.source_digest())" > $O/source_digest.txt
timeout 60 profiles/experiments/mfma_f32_probe > $O/mfma_f32_probe.log 2>&1; cat $O/mfma_f32_probe.log
timeout 120 profiles/experiments/xcd_probe > $O/xcd_probe.log 2>&1; cat $O/xcd_probe.log
timeout 1200 python -m pytest tests/test_gemm_qi.py tests/test_prefill_exact.py tests/test_dec_model.py tests/test_dec_engine.py tests/test_llama_runner.py tests/test_distributed.py tests/test_moe.py tests/test_zz_moe_prefill.py -q -m gpu -rf > $O/pytest_sel.log 2>&1; tail -15 $O/pytest_sel.log | cut -c1-300
(timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 1500 python bench.py 2>&1 | tail -1) > $O/bench_default.log; cut -c1-300 $O/bench_default.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
timeout 300 python profiles/experiments/gemm_lib_probe.py > $O/gemm_lib_probe.log 2>&1; cat $O/gemm_lib_probe.log
