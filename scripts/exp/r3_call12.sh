#!/bin/bash
cd /root/repo
bash scripts/profile_round.sh round3
O=gpurun_out/round3
(timeout 600 python bench.py --model mixtral --no-cpu-baseline 2>&1 | tail -1) > $O/bench_mixtral.log
cut -c1-1500 $O/bench_mixtral.log
