#!/bin/bash
# one short gpurun call: the GPU tests of the files given as arguments (scratch helper)
cd /root/repo
timeout 900 python -m pytest "$@" -q -m gpu 2>&1 | tail -15 | cut -c1-300
