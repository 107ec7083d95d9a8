#!/bin/bash
O=gpurun_out/r3c5; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
for a in 1 0; do
MRS_DEC_ATTN2=$a timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$a -o r -- python bench.py --no-cpu-baseline --steps 64 > $O/kt$a.log 2>&1
python scripts/rocprof_summary.py $(find $O/kt$a -name "*kernel_trace.csv" | head -1) --top 12 --match mrs:: | head -12 | cut -c1-200 | tee -a $O/stats.log
done
