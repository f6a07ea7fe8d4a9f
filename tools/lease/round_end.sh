#!/bin/bash
# round-end evidence (tools/profile_round.sh) + one full GPU pass:  bash tools/lease/round_end.sh [run name, default r6_end]
RUN=${1:-r6_end}
R=/root/repo/gpurun_out/$RUN
mkdir -p $R
cd /root/repo
timeout 2400 bash tools/profile_round.sh $RUN > $R/profile_round.log 2>&1
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_1.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_1.txt
tail -3 $R/pytest_1.txt | grep -v "^$"
cp gpurun_out/gpu_health.txt $R/gpu_health.txt 2>/dev/null
tail -c 700 $R/bench_full_1.json
