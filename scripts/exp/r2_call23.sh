#!/bin/bash
# call 23: PMC passes on the gate/up prefill GEMM (T = 512), both kernel variants
export TMPDIR=/tmp
O=gpurun_out/r2c23; mkdir -p $O
cd /root/repo
for v in 0 1; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    MRS_GEMM_VARIANT=$v timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/v${v}_p$i -o r -- python scripts/exp/gemm_one.py 512 28672 4096 > $O/v${v}_p$i.log 2>&1
    tail -2 $O/v${v}_p$i.log | cut -c1-200
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r2c23/v*_p*/')):
    for f in glob.glob(d + '*/*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_q' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print(d, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
