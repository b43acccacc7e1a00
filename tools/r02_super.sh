#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in long25 nedm noptrex; do
for f in 0 4096; do
  echo "== $w flags $f" >> $O/workloads.txt
  timeout -k 10 120 python3 tools/workload.py $w --debug-flags $f >> $O/workloads.txt 2>&1 || echo "FAILED rc=$?" >> $O/workloads.txt
done; done
grep -v amdgpu.ids $O/workloads.txt | cut -c1-330
