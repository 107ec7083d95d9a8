#!/bin/bash
# round 6, call A: the state of HEAD (round-5 kernels + bench.py fixes) -- GPU suite, default bench line (live roofline, batched leg), smoke, in-graph kernel trace,
# PMC traffic passes, and the upper bound of a producer-side activation image at batch 1 (bench_dec.py --img: the GEMV phases on a ready image)
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
python -c "import bench; print(bench.source_digest())" > $O/source_digest.txt
timeout 1500 python -m pytest tests -q -m gpu -x -rf > $O/pytest_gpu_full.log 2>&1; tail -3 $O/pytest_gpu_full.log
(timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 1200 python bench.py 2>&1 | tail -1) > $O/bench_default.log; cut -c1-400 $O/bench_default.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/write.log 2>&1
timeout 300 python scripts/bench_dec.py --b 1 > $O/dec_b1.log 2>&1; tail -8 $O/dec_b1.log
timeout 300 python scripts/bench_dec.py --b 1 --img > $O/dec_b1_img.log 2>&1; tail -8 $O/dec_b1_img.log
timeout 300 python scripts/bench_dec.py --b 1 --timeline > $O/dec_b1_tl.log 2>&1; tail -12 $O/dec_b1_tl.log
timeout 300 python scripts/bench_dec.py --b 1 --img --timeline > $O/dec_b1_img_tl.log 2>&1; tail -12 $O/dec_b1_img_tl.log
find $O -name "*.csv" | head
