#!/bin/bash
# Round-2 baseline on one GPU box: GPU tests, bench line, off-headline profiles, host-path and in-process HDF5 benches.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-1500
gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__ \
  && timeout -k 10 300 /tmp/host_path_bench > $O/r02_host_path_bench.txt 2>&1; cat $O/r02_host_path_bench.txt
HDF5=${HDF5_DIR:-/opt/conda}
gcc -O2 tools/h5_filter_bench.c -o /tmp/h5_filter_bench -I$HDF5/include -L$HDF5/lib -lhdf5 -Wl,-rpath,$HDF5/lib \
  && HDF5_PLUGIN_PATH=$R/deltarice_amd/plugin timeout -k 10 300 /tmp/h5_filter_bench /dev/shm/drx_bench.h5 > $O/r02_h5_filter_bench.txt 2>&1; cat $O/r02_h5_filter_bench.txt; rm -f /dev/shm/drx_bench.h5
for w in nab1 small20 small100; do timeout -k 10 200 python3 tools/workload.py $w >> $O/r02_small_batches.txt 2>&1; done; cat $O/r02_small_batches.txt
tools/profile_workloads.sh r02 config5 long25 nedm noptrex 2>&1 | tail -40
