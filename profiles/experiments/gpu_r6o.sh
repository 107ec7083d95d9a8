#!/bin/bash
# round 6, call O: Mixtral-shaped decode with the split router (MRS_MOE_ROUTER_SPLIT=1 default) vs the one-workgroup router, and its in-graph kernel trace
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
run() { name=$1; shift; (timeout 900 python bench.py --no-cpu-baseline --no-dropin --no-extra "$@" 2>&1 | tail -1) > $O/line_$name.log; python - "$O/line_$name.log" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "tok/s", j["value"], "ms", j["ms_per_step"], "step_frac", j.get("step_roofline_frac"), "ttft", j.get("ttft_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
run mixtral_split --model mixtral --steps 64
MRS_MOE_ROUTER_SPLIT=0 run mixtral_onewg --model mixtral --steps 64
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin --model mixtral --steps 64 > $O/kt.log 2>&1
python scripts/rocprof_summary.py $O/kt 2>/dev/null | head -30 || true
ls $O/kt | head; find $O/kt -name "*kernel_stats.csv" | head -2
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
