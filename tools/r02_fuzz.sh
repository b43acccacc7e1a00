#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02z; mkdir -p $O; cd $R
for seed in 101 202 303; do
  DRX_FUZZ_LOG=$O/fuzz_$seed.log timeout -k 10 280 python3 tests/fuzz_parity.py 250 $seed > $O/out_$seed.txt 2>&1; echo "seed $seed rc=$?"; tail -3 $O/out_$seed.txt
done
DRX_FUZZ_SCALE=6 DRX_FUZZ_LOG=$O/fuzz_big.log timeout -k 10 280 python3 tests/fuzz_parity.py 60 404 > $O/out_big.txt 2>&1; echo "big rc=$?"; tail -3 $O/out_big.txt
