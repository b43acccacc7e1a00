#!/bin/bash
# block decoder: parity subset, then the small / long workloads with the new and the old path
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "few or random_vs_oracle or golden_batch or corrupt" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
for w in small20 small100 nab1 nedm noptrex long25; do
  for f in 0 1024; do
    echo "== $w flags $f" >> $O/workloads.txt
    timeout -k 10 90 python3 tools/workload.py $w --debug-flags $f >> $O/workloads.txt 2>&1 || echo "FAILED rc=$?" >> $O/workloads.txt
  done
done
grep -v amdgpu.ids $O/workloads.txt | cut -c1-420
