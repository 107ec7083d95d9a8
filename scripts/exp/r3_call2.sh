#!/bin/bash
O=gpurun_out/r3c2; mkdir -p $O
cd /root/repo
MRS_EXT_LIB=libmrs_hip_ext_tl.so timeout 600 python scripts/exp/timeline.py 2>&1 | tail -40 | tee $O/timeline.log
