#!/bin/bash
# Round 4, verdict item 1: what the deferred-placement encoder may pay in occupancy.  k_encode_fused WITHOUT its look-back
# (-DDRX_ABLATION, flag 128: positions are wrong, results invalid) and in full (flag 0) at 8 ... 16 wavefronts per CU:
# DRX_ENC_WAVES wavefronts per workgroup x DRX_ENC_LDS_PAD bytes of dynamic LDS decide how many workgroups a CU holds.
# usage (GPU box): tools/r04_enc_occupancy.sh   -> gpurun_out/r04_enc_occ/table.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_enc_occ; mkdir -p $O; cd $R; : > $O/table.txt
run() {  # lib pad label
  echo "== $3" | tee -a $O/table.txt
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$1.so DRX_ENC_LDS_PAD=$2 timeout -k 10 200 python3 tools/enc_only.py 128 0 128 0 2>>$O/err.txt | tee -a $O/table.txt
}
run abl8 0      "8 waveforms x 8.2 KB, 2 workgroups = 16 wavefronts per CU"
run abl8 20000  "8 waveforms x 8.2 KB + 20 KB, 1 workgroup = 8 per CU"
run abl4 0      "4 x 8.2 KB, 4 workgroups = 16 per CU"
run abl4 12000  "4 x 8.2 KB + 12 KB, 3 workgroups = 12 per CU"
run abl4 24000  "4 x 8.2 KB + 24 KB, 2 workgroups = 8 per CU"
run abl12 0     "12 x 8.2 KB, 1 workgroup = 12 per CU"
run abl16 0     "16 x 8.2 KB, 1 workgroup = 16 per CU"
