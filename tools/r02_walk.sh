#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02w; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walk or ragged or corrupt or random_vs_oracle or pieces" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for f in 0; do
  echo "== config5 flags $f" >> $O/workloads.txt
  timeout -k 10 120 python3 tools/workload.py config5 --debug-flags $f >> $O/workloads.txt 2>&1 || echo "FAILED rc=$?" >> $O/workloads.txt
done
grep -v amdgpu.ids $O/workloads.txt | cut -c150-420
for ch in 100 25; do
for f in 0; do
  echo "== len_sweep $ch chunks, flags $f" >> $O/sweep.txt
  DRX_SWEEP_CHUNKS=$ch DRX_DEBUG_FLAGS=$f timeout -k 10 300 python3 tools/len_sweep.py 64 128 512 1024 2048 3072 >> $O/sweep.txt 2>&1 || echo "FAILED rc=$?" >> $O/sweep.txt
done; done
grep -v amdgpu.ids $O/sweep.txt
