#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py -m gpu -x -q 2>&1 | tail -15
