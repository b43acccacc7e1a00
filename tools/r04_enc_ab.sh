#!/bin/bash
# k_encode_stream variants, interleaved: tools/r04_enc_ab.sh TAG "name:lib:impl:flags" ...   (lib "-" = the product library)
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift; O=$R/gpurun_out/r04_encab_$TAG; mkdir -p $O; cd $R; : > $O/ab.txt
for rep in 1 2 3; do for spec in "$@"; do
  IFS=: read name lib impl flags <<< "$spec"
  if [ "$lib" = "-" ]; then unset DRX_LIB_PATH; else export DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$lib.so; fi
  DRX_ENCODE_IMPL=$impl timeout -k 10 200 python3 tools/enc_only.py ${flags:-0} 2>&1 | grep -E "stamps|flags|scanner" | tail -3 | sed "s/^/$name: /" | tee -a $O/ab.txt
done; done
