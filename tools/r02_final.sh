#!/bin/bash
# end-of-round evidence on one box: GPU tests, bench line, off-headline profiles, host path, in-process HDF5, SQ counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
bash tools/refresh_profiles.sh r02 2>&1 | tail -12; cp gpurun_out/refresh/r02_* $O/ 2>/dev/null
gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__ \
  && timeout -k 10 300 /tmp/host_path_bench > $O/r02_host_path_bench_final.txt 2>&1; cat $O/r02_host_path_bench_final.txt
HDF5=${HDF5_DIR:-/opt/conda}
gcc -O2 tools/h5_filter_bench.c -o /tmp/h5_filter_bench -I$HDF5/include -L$HDF5/lib -lhdf5 -Wl,-rpath,$HDF5/lib \
  && HDF5_PLUGIN_PATH=$R/deltarice_amd/plugin timeout -k 10 300 /tmp/h5_filter_bench /dev/shm/drx_bench.h5 > $O/r02_h5_filter_bench_final.txt 2>&1; cat $O/r02_h5_filter_bench_final.txt; rm -f /dev/shm/drx_bench.h5
for w in nab1 small20 small100; do timeout -k 10 200 python3 tools/workload.py $w 2>/dev/null >> $O/r02_small_batches_final.txt; done; cut -c1-400 $O/r02_small_batches_final.txt
tools/profile_workloads.sh r02f config5 long25 nedm noptrex 2>&1 | grep -E '^\{|k_decode|k_seg|k_bw|k_pw' | cut -c1-330
timeout -k 10 500 python3 tools/bench_configs.py 2>/dev/null > $O/r02_bench_configs.txt; cat $O/r02_bench_configs.txt
timeout -k 10 300 python3 tools/h5_direct_bench.py 2>/dev/null > $O/r02_h5_direct_bench.txt; cat $O/r02_h5_direct_bench.txt
for ch in 25 100; do DRX_SWEEP_CHUNKS=$ch timeout -k 10 400 python3 tools/len_sweep.py 64 128 512 1024 2048 3072 3500 5000 7000 9000 12000 16384 32768 65536 81920 500000 2>/dev/null > $O/r02_len_sweep_${ch}chunks.txt; cat $O/r02_len_sweep_${ch}chunks.txt; done
