#!/bin/bash
# Evidence pass of a round in one gpurun call: scripts/profile_round.sh (GPU suite, default bench line, smoke, rocprofv3 kernel trace + PMC passes), then the
# other bench lines of profiles/round<N>_bench_lines.md.
cd /root/repo
export TMPDIR=/tmp
bash scripts/profile_round.sh round3
O=gpurun_out/round3
run() { name=$1; shift; (timeout 500 python bench.py --no-cpu-baseline --no-dropin "$@" 2>&1 | tail -1) > $O/line_$name.log; python - "$O/line_$name.log" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print(sys.argv[2], "tok/s", j["value"], "ms", j["ms_per_step"], "step_frac", j.get("step_roofline_frac"), "kernel", r.get("us_per_launch"), "us", r.get("frac"), "ttft", j.get("ttft_ms"), "prefill_frac", j.get("prefill_roofline", {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run mixtral --model mixtral
run q8isq --quant q8_0_isq --steps 128
run batch8 --batch 8 --steps 128
run p2048 --prompt-len 2048 --steps 128
run 70b --model 70b --steps 32
