#!/bin/bash
# Round 4's evidence set.  Part A: the headline (bench line with its own PMC passes, the same command under rocprofv3 kernel
# stats, the two PMC passes), what the in-run PMC passes cost the headline (advisor), BASELINE configs 3 and 5.  Part B: the
# off-headline workloads (HIP-event JSON, kernel stats, PMC traffic), the H5Z callback from C.
# usage (GPU box, repo root): tools/r04_final.sh A|B      -> gpurun_out/r04f/ and gpurun_out/refresh/
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r04f; mkdir -p $O
if [ "$1" = "A" ]; then
  # the headline without and with the in-run PMC passes, twice each, before any profiler has run in this session
  for i in 1 2; do
    timeout -k 10 200 python3 bench.py --no-collect --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('no-collect  value %.1f GB/s  encode %.3f ms  decode %.3f ms' % (d['value'], d['kernel_ms']['encode_kernel'], d['kernel_ms']['decode_kernel']))"
    timeout -k 10 300 python3 bench.py --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('collect     value %.1f GB/s  encode %.3f ms  decode %.3f ms  (traffic collected in run: %s)' % (d['value'], d['kernel_ms']['encode_kernel'], d['kernel_ms']['decode_kernel'], d['roofline']['traffic_collected_in_run']))"
  done > $O/r04_collect_effect.txt 2>&1
  cat $O/r04_collect_effect.txt
  tools/refresh_profiles.sh r04 > $O/refresh.log 2>&1; tail -12 $O/refresh.log | cut -c1-600
  timeout -k 10 500 python3 tools/bench_configs.py > $O/r04_bench_configs.txt 2>&1; cat $O/r04_bench_configs.txt
else
  gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__ && /tmp/host_path_bench > $O/r04_host_path_bench.txt 2>&1
  tail -4 $O/r04_host_path_bench.txt
  rm -f $O/r04_more_workloads.txt
  for w in "config5 --sideband" "nab1" "nab1 --sideband" "small20" "small100" "nab100"; do
    timeout -k 10 200 python3 tools/workload.py $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-24s path %2d ratio %.4f decode %.3f ms (walk %.3f) frac %.3f | encode %.3f ms frac %.3f' % ('$w', d['decode_path'], d['ratio'], d['decode_ms']['total'], d['decode_ms']['walk'], d['decode_frac_of_8TBps'], d['encode_ms']['total'], d['encode_frac_of_8TBps']))" >> $O/r04_more_workloads.txt
  done
  cat $O/r04_more_workloads.txt
  tools/profile_workloads.sh r04f config5 long25 nedm noptrex noptrex_fir4 nedm_fir4 raglong raglong_fir4 > $O/workloads.log 2>&1
  grep -h "decode_frac" $O/r04f_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-14s decode %.3f ms frac %.3f | encode %.3f ms frac %.3f' % (d['workload'], d['decode_ms']['total'], d['decode_frac_of_8TBps'], d['encode_ms']['total'], d['encode_frac_of_8TBps']))"
fi
