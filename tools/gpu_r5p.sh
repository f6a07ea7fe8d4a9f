#!/bin/bash
# round 5, last step: round-end evidence (tools/profile_round.sh) + two full GPU passes
R=/root/repo/gpurun_out/r5_z
mkdir -p $R
cd /root/repo
bash tools/profile_round.sh r5_z > $R/profile_round.log 2>&1
cd /root/repo
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_$i.txt 2>&1
  echo "pytest rc=$?" >> $R/pytest_$i.txt
  tail -3 $R/pytest_$i.txt | grep -v "^$"
done
cp gpurun_out/gpu_health.txt $R/gpu_health.txt 2>/dev/null
tail -3 $R/bench_full_1.json | cut -c1-600
