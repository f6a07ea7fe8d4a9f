#!/bin/bash
# round 5, GPU pass B: CAS switch, self-test, diagnosing conftest; then a full pass
R=/root/repo/gpurun_out/r5_b
mkdir -p $R
cd /root/repo
timeout 300 python -m pytest tests/_diag_demo_gpu.py -q > $R/diag_demo.txt 2>&1
grep -n "diagnosis\|same test body\|self-test\|last accumulating\|BAD" $R/diag_demo.txt | head -20
timeout 1200 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -12 $R/pytest.txt
