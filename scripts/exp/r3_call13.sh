#!/bin/bash
O=gpurun_out/r3c13; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -x -v -m gpu > $O/pytest_a.log 2>&1
tail -40 $O/pytest_a.log | cut -c1-300
dmesg 2>/dev/null | tail -5
echo "== pmc with block weights"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python bench.py --no-cpu-baseline --no-dropin --weights blocks --steps 16 > $O/fetch.log 2>&1
tail -5 $O/fetch.log | cut -c1-300
echo "== pmc gaussian no dropin"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch2 -o r -- python bench.py --no-cpu-baseline --no-dropin --steps 16 > $O/fetch2.log 2>&1
tail -5 $O/fetch2.log | cut -c1-300
