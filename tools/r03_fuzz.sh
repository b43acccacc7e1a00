#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
for seed in 2101 2202 2303 2404; do
  DRX_FUZZ_LOG=$O/fuzz_$seed.log timeout -k 10 280 python3 tests/fuzz_parity.py 250 $seed > $O/out_$seed.txt 2>&1; echo "seed $seed rc=$?"; tail -3 $O/out_$seed.txt
done
DRX_FUZZ_SCALE=6 DRX_FUZZ_LOG=$O/fuzz_big.log timeout -k 10 280 python3 tests/fuzz_parity.py 80 2505 > $O/out_big.txt 2>&1; echo "big rc=$?"; tail -3 $O/out_big.txt
DRX_FUZZ_CORRUPT=1 DRX_FUZZ_LOG=$O/fuzz_corrupt.log timeout -k 10 280 python3 tests/fuzz_parity.py 200 2606 > $O/out_corrupt.txt 2>&1; echo "corrupt rc=$?"; tail -3 $O/out_corrupt.txt
