#!/bin/bash
O=gpurun_out/r3c14; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -x -q -m gpu -k "not 8b and not prefill_512 and not sliding" > $O/loop$i.log 2>&1
  echo "loop $i rc=$?"; tail -3 $O/loop$i.log | cut -c1-200
  grep -i "fault\|abort\|error" $O/loop$i.log | head -5
done
OUT=gpurun_out/round3
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python bench.py --no-cpu-baseline --no-dropin --steps 32 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python bench.py --no-cpu-baseline --no-dropin --steps 32 > $OUT/write.log 2>&1
ls -la $OUT/fetch $OUT/write
