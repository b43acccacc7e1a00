#!/bin/bash
# first runs of k_encode_stream: parity, then timing against k_encode_fused, then the stamps of both
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_first; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or golden_batch or random_vs_oracle or general_prediction or capacity" > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
for rep in 1 2; do for impl in 2 1; do
  DRX_ENCODE_IMPL=$impl timeout -k 10 200 python3 tools/enc_only.py 0 2>&1 | tail -1 | sed "s/^/impl $impl: /"
done; done | tee $O/timing.txt
for impl in 2 1; do
  DRX_ENCODE_IMPL=$impl DRX_LIB_PATH=$R/deltarice_amd/variants/lib_encstamps.so timeout -k 10 200 python3 tools/enc_only.py 0 2>&1 | grep -E "stamps|flags" | tail -3
done | tee $O/stamps.txt
