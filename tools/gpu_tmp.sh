#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python tools/rgcn_grouped_probe.py 50 15,10 2>&1 | grep "grouped="
timeout 300 python tools/rgcn_grouped_probe.py 30 15,10 256 2>&1 | grep "grouped=True"
timeout 300 python tools/rgcn_grouped_probe.py 30 15,10 128 f32 2>&1 | grep "grouped=True"
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py -m gpu -x -q 2>&1 | tail -2
