#!/bin/bash
# diagnostic: per-phase times and timelines of the batch-8 GEMV launches on a pre-built activation image (bench_dec.py --img)
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
timeout 120 python scripts/bench_dec.py --b 8 --img --reps 8 --phases qkv,o,gate_up,down4 > $O/dec8.log 2>&1; grep phase $O/dec8.log | cut -c1-130
timeout 120 python scripts/bench_dec.py --b 8 --img --timeline --phases qkv,o,gate_up,down4 > $O/tl8.log 2>&1; grep median $O/tl8.log | cut -c1-300
