#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
for ch in 2 10 25; do for f in 0 256; do
  echo "== chunks $ch flags $f" >> $O/sweep.txt
  DRX_SWEEP_CHUNKS=$ch DRX_DEBUG_FLAGS=$f timeout -k 10 120 python3 tools/len_sweep.py 4096 7000 16384 65536 2>&1 | grep -v amdgpu >> $O/sweep.txt
done; done
cat $O/sweep.txt
gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__ && timeout -k 10 120 /tmp/host_path_bench | tee $O/host_path.txt
