#!/bin/bash
# full GPU pass + bench line
R=/root/repo/gpurun_out/r6_full
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_full.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_full.txt
tail -4 $R/pytest_full.txt
timeout 900 python bench.py > $R/bench.json 2> $R/bench.err
tail -c 1500 $R/bench.json
