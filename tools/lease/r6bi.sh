#!/bin/bash
R=/root/repo/gpurun_out/r6_bi
mkdir -p $R
cd /root/repo
PYTHONPATH=/root/repo python tools/rgcn_item_balance.py 768 2>&1 | grep -v amdgpu.ids | tee $R/balance.txt
