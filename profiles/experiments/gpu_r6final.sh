#!/bin/bash
# round 6, closing evidence in one call: scripts/profile_round.sh (full GPU suite, default bench line, smoke, in-graph kernel trace, FETCH / WRITE_SIZE passes of THIS build),
# the other bench lines of profiles/round6_bench_lines.md, and a PMC pass over the exact prompt GEMM
cd /root/repo
export TMPDIR=/tmp
bash scripts/profile_round.sh r6final
O=gpurun_out/r6final
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
run p2048 --prompt-len 2048 --steps 128
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc_qi -o r -- python scripts/bench_gemm.py --qi --types q4_k --t 512 > $O/pmc_qi.log 2>&1
tail -5 $O/pmc_qi.log | cut -c1-200
