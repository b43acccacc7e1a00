#!/bin/bash
# direct-chunk file <-> VRAM path, pipelined (round 4): 20 000 x 7000 and 200 000 x 7000, raw preads and H5Dread_chunk
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; : > $O/r04_h5_direct_bench.txt
for rows in 20000 200000; do
  echo "== $rows x 7000 (file -> VRAM by four threads of pread)" | tee -a $O/r04_h5_direct_bench.txt
  timeout -k 10 300 python3 tools/h5_direct_bench.py $rows 7000 2000 2>&1 | tail -5 | tee -a $O/r04_h5_direct_bench.txt
  echo "== $rows x 7000 (DRX_H5_NO_RAW=1: file -> VRAM by H5Dread_chunk)" | tee -a $O/r04_h5_direct_bench.txt
  DRX_H5_NO_RAW=1 timeout -k 10 300 python3 tools/h5_direct_bench.py $rows 7000 2000 2>&1 | grep "file->VRAM" | head -1 | tee -a $O/r04_h5_direct_bench.txt
done
