#!/bin/bash
# round 6, closing evidence of the second session in one call: scripts/profile_round.sh (full GPU suite, default bench line, smoke, in-graph kernel trace, FETCH / WRITE_SIZE passes of
# THIS build), the other bench lines of profiles/round6_bench_lines.md, the batched lines on both routes, and the kernel trace of a batch-8 step
cd /root/repo
export TMPDIR=/tmp
bash scripts/profile_round.sh r6final3
O=gpurun_out/r6final3
run() { name=$1; shift; (timeout 700 python bench.py --no-cpu-baseline --no-dropin --no-extra "$@" 2>&1 | tail -1) > $O/line_$name.log; python - "$O/line_$name.log" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print(sys.argv[2], "tok/s", j["value"], "ms", j["ms_per_step"], "step_frac", j.get("step_roofline_frac"), "kernel", r.get("us_per_launch"), "us", r.get("frac"), "ttft", j.get("ttft_ms"),
          "prefill_frac", j.get("prefill_roofline", {}).get("frac"), "bf16", (j.get("prefill_bf16") or {}).get("frac"), "bf16_fused", ((j.get("prefill_bf16") or {}).get("fused_dequant_kernels") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run mixtral --model mixtral --steps 128
run q8isq --quant q8_0_isq --steps 128
run batch8 --batch 8 --steps 128
run batch4 --batch 4 --steps 128
run batch3 --batch 3 --steps 128
run batch2 --batch 2 --steps 128
MRS_DEC_MM=0 run batch8_valu --batch 8 --steps 128
MRS_DEC_MM=0 run batch4_valu --batch 4 --steps 128
run p2048 --prompt-len 2048 --steps 128
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b8 -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin --batch 8 --steps 64 > $O/kt_b8.log 2>&1
f=$(find $O/kt_b8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 "$f"; grep -E "dec_mm|dec_act_image|dec_attn2|sample|embedding|dec_gemv" "$f" | head -20) | cut -c1-220 > $O/kt_b8_stats.txt; cat $O/kt_b8_stats.txt
find $O/kt_b8 -name "*kernel_trace.csv" -delete
