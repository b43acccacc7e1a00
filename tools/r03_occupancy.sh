#!/bin/bash
# Round 3, verdict item 1(a): what waves per CU buy k_decode_lanes at an UNCHANGED instruction stream.
#   lib_abl     = production kernels + ablation switches (-DDRX_ABLATION); DRX_DEC_LDS_PAD adds dynamic LDS per wavefront
#   lib_noobuf  = the same with the transposition buffer and the write-out compiled away (-DDRX_DEC_NOOBUF: 17.9 KB, 9 per CU)
# usage (GPU box): tools/r03_occupancy.sh      -> gpurun_out/r03_occ/table.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_occ; mkdir -p $O; cd $R; : > $O/table.txt
run() {  # lib pad flags label
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$1.so DRX_DEC_LDS_PAD=$2 timeout -k 10 150 python3 bench.py --no-collect --cpu-seconds 0 --steps 6 --warmup 2 --debug-flags $3 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s dec %.3f ms  enc %.3f ms' % ('$4', d['kernel_ms']['decode_kernel'], d['kernel_ms']['encode_kernel']))" | tee -a $O/table.txt
}
for rep in 1 2; do
run abl 0 0       "full kernel, 26.1 KB, 6 waves/CU"
run abl 5000 0    "full kernel, 31.1 KB, 5 waves/CU"
run abl 13000 0   "full kernel, 39.1 KB, 4 waves/CU"
run abl 27000 0   "full kernel, 53.1 KB, 3 waves/CU"
run abl 0 1       "no output stores, 6 waves/CU"
run abl 0 2       "no stream loads, 6 waves/CU"
run abl 0 3       "no loads, no stores, 6 waves/CU"
run noobuf 0 3      "noobuf, no loads/stores, 17.9 KB, 9/CU"
run noobuf 2048 3   "noobuf, no loads/stores, 20.0 KB, 8/CU"
run noobuf 4608 3   "noobuf, no loads/stores, 22.5 KB, 7/CU"
run noobuf 8192 3   "noobuf, no loads/stores, 26.1 KB, 6/CU"
run noobuf 13312 3  "noobuf, no loads/stores, 31.2 KB, 5/CU"
run noobuf 22000 3  "noobuf, no loads/stores, 39.9 KB, 4/CU"
run noobuf 0 1      "noobuf, loads kept, 17.9 KB, 9/CU"
run noobuf 4608 1   "noobuf, loads kept, 22.5 KB, 7/CU"
run noobuf 8192 1   "noobuf, loads kept, 26.1 KB, 6/CU"
done
