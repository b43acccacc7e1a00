#!/bin/bash
# end-of-round evidence on one box.  Order matters: everything that is TIMED runs before the first profiler pass (the
# rocprofv3 --pmc passes leave the device / the host runtime in a state in which later timings of the same session come out
# slower -- the host path is run once more at the very end to show it).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
HDF5=${HDF5_DIR:-/opt/conda}
gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__
gcc -O2 tools/h5_filter_bench.c -o /tmp/h5_filter_bench -I$HDF5/include -L$HDF5/lib -lhdf5 -Wl,-rpath,$HDF5/lib
# ---- A: timings ----
timeout -k 10 300 /tmp/host_path_bench > $O/r02_host_path_bench_final.txt 2>&1; cat $O/r02_host_path_bench_final.txt
HDF5_PLUGIN_PATH=$R/deltarice_amd/plugin timeout -k 10 300 /tmp/h5_filter_bench /dev/shm/drx_bench.h5 > $O/r02_h5_filter_bench_final.txt 2>&1; cat $O/r02_h5_filter_bench_final.txt; rm -f /dev/shm/drx_bench.h5
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/r02_bench.json; cut -c1-600 $O/r02_bench.json
for w in nab1 small20 small100; do timeout -k 10 200 python3 tools/workload.py $w 2>/dev/null >> $O/r02_small_batches_final.txt; done; cut -c1-400 $O/r02_small_batches_final.txt
for w in config5 long25 nedm noptrex; do timeout -k 10 300 python3 tools/workload.py $w 2>/dev/null > $O/r02f_$w.json; cut -c1-400 $O/r02f_$w.json; done
timeout -k 10 500 python3 tools/bench_configs.py 2>/dev/null > $O/r02_bench_configs.txt; cat $O/r02_bench_configs.txt
timeout -k 10 300 python3 tools/h5_direct_bench.py 2>/dev/null > $O/r02_h5_direct_bench.txt; cat $O/r02_h5_direct_bench.txt
for ch in 25 100; do DRX_SWEEP_CHUNKS=$ch timeout -k 10 400 python3 tools/len_sweep.py 64 128 512 1024 2048 3072 3500 5000 7000 9000 12000 16384 32768 65536 81920 500000 2>/dev/null > $O/r02_len_sweep_${ch}chunks.txt; cat $O/r02_len_sweep_${ch}chunks.txt; done
# ---- B: the GPU tests ----
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
# ---- C: profiler passes (kernel stats, PMC traffic) ----
SKIP_PLAIN=1 bash tools/refresh_profiles.sh r02 2>&1 | tail -12; cp gpurun_out/refresh/r02_* $O/ 2>/dev/null
SKIP_PLAIN=1 tools/profile_workloads.sh r02f config5 long25 nedm noptrex 2>&1 | grep -E 'k_decode|k_seg|k_bw|k_pw|k_encode' | cut -c1-200
# ---- D: the host path once more, after the profilers ----
timeout -k 10 300 /tmp/host_path_bench > $O/r02_host_path_bench_after_profilers.txt 2>&1; cat $O/r02_host_path_bench_after_profilers.txt
