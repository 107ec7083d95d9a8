#!/bin/bash
# batched decode: the activation image built once per phase (mrs_dec_act_image + *_img): parity tests, then batch 2 / 4 / 8 with the threshold off / 2 / 3
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout 600 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -m gpu -x -q -k "image or batch" > $O/t1.log 2>&1; tail -2 $O/t1.log
for thr in 99 2 3; do for b in 2 4 8; do
  if [ $thr = 3 ] && [ $b != 2 ] ; then continue; fi
  MRS_DEC_IMG_MIN_B=$thr timeout 300 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 --batch $b > $O/bench_t${thr}_b$b.log 2>&1; echo "thr $thr b $b: $(tail -1 $O/bench_t${thr}_b$b.log | cut -c1-110)"
done; done
MRS_DEC_IMG_MIN_B=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 --batch 8 > $O/kt8.log 2>&1
f=$(find $O/kt8 -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py $f --top 10 --match dec 2>&1 | cut -c1-200
