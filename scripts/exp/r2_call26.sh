#!/bin/bash
O=gpurun_out/r2c26; mkdir -p $O
cd /root/repo
{
for v in 0 1; do for t in 512 2048; do
echo "== variant $v T=$t"; MRS_GEMM_VARIANT=$v timeout 300 python scripts/bench_gemm.py --big --t $t --types q5_k,q8_0
done; done
} 2>&1 | grep -v amdgpu.ids | tee $O/gemm.log | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('  %-5s %-8s %7.1f us %6.1f TF' % (j['type'], j['shape'], j['us'], j['TFLOPs']))
    else: print(l)
"
