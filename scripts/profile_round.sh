#!/bin/bash
# Evidence run for one round (on the GPU box, from the repo root): GPU tests, the default bench line, smoke, then the rocprofv3 passes
# that profiles/*.md is generated from (scripts/make_profile_summary.py).  Counter passes are separate runs with --kernel-trace only.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}
mkdir -p $OUT
python -c "import bench; print(bench.source_digest())" > $OUT/source_digest.txt
timeout 1800 python -m pytest tests -q -m gpu -rf > $OUT/pytest_gpu_full.log 2>&1
(grep -E "passed|failed|error|FAILED|Fatal|fault" $OUT/pytest_gpu_full.log | tail -12) > $OUT/pytest_gpu.log
(timeout 900 python bench.py 2>&1 | tail -1) > $OUT/bench_default.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $OUT/smoke.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python bench.py --no-cpu-baseline --no-extra > $OUT/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $OUT/write.log 2>&1
tail -1 $OUT/kt.log | cut -c1-300
cat $OUT/pytest_gpu.log $OUT/smoke.log
cut -c1-1200 $OUT/bench_default.log
find $OUT -name "*.csv" | head -20
