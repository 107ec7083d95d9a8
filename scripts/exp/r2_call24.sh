#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2c24; mkdir -p $O
cd /root/repo
{
echo "== variant 1 (wave-specialised, 32-weight producer units), T=512"; MRS_GEMM_VARIANT=1 timeout 300 python scripts/bench_gemm.py --big --t 512 --types q4_k,q6_k
echo "== variant 1, T=2048"; MRS_GEMM_VARIANT=1 timeout 300 python scripts/bench_gemm.py --big --t 2048 --types q4_k
} > $O/gemm.log 2>&1
cat $O/gemm.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gemm.py -m gpu -x -q 2>&1 | tail -2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  MRS_GEMM_VARIANT=1 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/v1_p$i -o r -- python scripts/exp/gemm_one.py 512 28672 4096 > $O/v1_p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r2c24/v*_p*/')):
    for f in glob.glob(d + '*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_q' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        kt = f.replace('counter_collection', 'kernel_trace')
        du = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt)) if 'gemm_q' in r['Kernel_Name']]
        print(d.split('/')[-2], 'us %.1f' % (sum(du)/len(du)), {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
