#!/bin/bash
# round 6, call C: full GPU suite on the tree with the bf16 shadow path, the prefetching MFMA prompt attention; default bench line; kernel trace
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
python -c "import bench; print(bench.source_digest())" > $O/source_digest.txt
timeout 300 python -m pytest tests/test_bf16_shadow.py tests/test_prefill_exact.py -q -m gpu -x -rf > $O/pytest_new.log 2>&1; tail -6 $O/pytest_new.log | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -rf > $O/pytest_gpu_full.log 2>&1; tail -12 $O/pytest_gpu_full.log | cut -c1-300
(timeout 1500 python bench.py 2>&1 | tail -1) > $O/bench_default.log; cut -c1-300 $O/bench_default.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
(timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --prompt-len 2048 --steps 64 2>&1 | tail -1) > $O/bench_p2048.log; cut -c1-300 $O/bench_p2048.log
