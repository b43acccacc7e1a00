#!/bin/bash
# headline encoder: where the time goes (ablation build; results invalid)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_enc; mkdir -p $O; cd $R; : > $O/table.txt
run() {  # lib flags label
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$1.so timeout -k 10 150 python3 bench.py --no-collect --cpu-seconds 0 --steps 6 --warmup 2 --debug-flags $2 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-50s enc %.3f ms' % ('$3', d['kernel_ms']['encode_kernel']))" | tee -a $O/table.txt
}
for rep in 1 2; do
run abl 1048576 "full (flag 1048576 = nothing for the encoder)"
run abl 32  "no emission"
run abl 64  "no copy-out"
run abl 128 "no look-back"
run abl 96  "no emission, no copy-out"
run abl 224 "no emission, no copy-out, no look-back"
run abl 16  "per-code emission"
done
