#!/bin/bash
# staged flush: where the time goes (ablation builds; results invalid)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_stg; mkdir -p $O; cd $R; : > $O/table.txt
run() {  # lib pad flags label
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$1.so DRX_DEC_LDS_PAD=$2 timeout -k 10 150 python3 bench.py --no-collect --cpu-seconds 0 --steps 6 --warmup 2 --debug-flags $3 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-50s dec %.3f ms' % ('$4', d['kernel_ms']['decode_kernel']))" | tee -a $O/table.txt
}
for rep in 1 2; do
run nw4a 0 1048576 "one-wave kernel (flag 1048576), 6/CU"
run nw4a 0 16      "nw4 full (dbg 16 = nothing)"
run nw4a 0 1       "nw4 no stores"
run nw4a 0 8       "nw4 no lock"
run nw4a 0 9       "nw4 no lock, no stores"
done
