#!/bin/bash
# pieces encoder: parity, then config5 / len_sweep with and without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or ragged or random_vs_oracle or golden_batch or capacity" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for f in 0 4096; do
  echo "== config5 flags $f" >> $O/workloads.txt
  timeout -k 10 120 python3 tools/workload.py config5 --debug-flags $f >> $O/workloads.txt 2>&1 || echo "FAILED rc=$?" >> $O/workloads.txt
done
grep -v amdgpu.ids $O/workloads.txt | cut -c1-420
for f in 0 4096; do
  echo "== len_sweep 100 chunks, flags $f" >> $O/sweep.txt
  DRX_SWEEP_CHUNKS=100 DRX_DEBUG_FLAGS=$f timeout -k 10 300 python3 tools/len_sweep.py 128 256 512 1024 2048 3072 3500 7000 9000 12000 16384 32768 65536 >> $O/sweep.txt 2>&1 || echo "FAILED rc=$?" >> $O/sweep.txt
done
grep -v amdgpu.ids $O/sweep.txt
