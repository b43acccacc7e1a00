#!/bin/bash
# Regenerates the artefacts under profiles/ from one GPU box: bench JSON, rocprofv3 kernel stats of the same
# command, and the two PMC passes for HBM traffic.  Run from the repo root on the GPU box (gpurun).
# usage: tools/refresh_profiles.sh [TAG]   (default r02; output in gpurun_out/refresh, to be copied into profiles/)
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
cd $R
# the plain bench line first: the profiler runs below (the --pmc passes in particular) may leave the device in another state
if [ -z "$SKIP_PLAIN" ]; then
  timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1
  tail -1 $O/bench.log > $O/${TAG}_bench.json
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --cpu-seconds 0 --no-collect > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $O/pmc_rd -o p --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-collect > $O/pmc_rd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_wr -o p --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-collect > $O/pmc_wr.log 2>&1
DRX_TRAFFIC_SOURCE="rocprofv3 --pmc, python3 bench.py --cpu-seconds 0 --steps 3 --warmup 1, one MI355X" python3 profiles/make_traffic_json.py profiles/${TAG}_pmc_traffic.json $O/pmc_rd/p_counter_collection.csv $O/pmc_wr/p_counter_collection.csv
python3 profiles/trim_stats.py $O/stats/s_kernel_stats.csv $O/${TAG}_kernel_stats.csv
grep '^{"metric"' $O/stats.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
cp profiles/${TAG}_pmc_traffic.json $O/${TAG}_pmc_traffic.json
rm -rf $O/stats $O/pmc_rd $O/pmc_wr
cat $O/${TAG}_kernel_stats.csv | head -8
[ -z "$SKIP_PLAIN" ] && tail -1 $O/bench.log | cut -c1-1500 || true
