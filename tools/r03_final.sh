#!/bin/bash
# Round 3's evidence set in one session: the bench line (with its own PMC passes), the same command under rocprofv3 kernel
# stats, the off-headline workloads (HIP-event JSON, kernel stats, PMC traffic), the H5Z callback from C, the GPU tests.
# usage (GPU box, repo root): tools/r03_final.sh      -> gpurun_out/r03f/ and gpurun_out/refresh/
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r03f; mkdir -p $O
gcc -O2 -Iinclude -I/opt/rocm/include tools/host_path_bench.c -o /tmp/host_path_bench -Ldeltarice_amd -ldeltarice_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/deltarice_amd -Wl,-rpath,/opt/rocm/lib -lm -D__HIP_PLATFORM_AMD__ && /tmp/host_path_bench > $O/r03_host_path_bench.txt 2>&1
tools/refresh_profiles.sh r03 > $O/refresh.log 2>&1
rm -f $O/r03_more_workloads.txt
(for w in nab100 noptrex nedm long25 config5; do tools/r03_noise.sh $w; done) > $O/r03_noise_sweep.txt 2>&1
tools/r03_quiet.sh > $O/r03_quiet_sweep.txt 2>&1
tools/r03_loud.sh > $O/r03_loud_sweep.txt 2>&1
for w in "config5 --sideband" "nab1" "nab1 --sideband" "small20" "small100" "noptrex_fir4 --white"; do
  timeout -k 10 200 python3 tools/workload.py $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-24s path %2d ratio %.4f decode %.3f ms (walk %.3f) frac %.3f | encode %.3f ms frac %.3f' % ('$w', d['decode_path'], d['ratio'], d['decode_ms']['total'], d['decode_ms']['walk'], d['decode_frac_of_8TBps'], d['encode_ms']['total'], d['encode_frac_of_8TBps']))" >> $O/r03_more_workloads.txt
done
tools/profile_workloads.sh r03f config5 long25 nedm noptrex noptrex_fir4 nedm_fir4 raglong raglong_fir4 > $O/workloads.log 2>&1
tail -3 $O/workloads.log
