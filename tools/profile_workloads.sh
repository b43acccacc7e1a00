#!/bin/bash
# rocprofv3 evidence for the off-headline workloads (tools/workload.py): kernel stats + the two PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, MI355X_MICROARCH.md), trimmed into gpurun_out/$TAG/ as
#   ${TAG}_${name}_kernel_stats.csv   ${TAG}_${name}_pmc_traffic.json   ${TAG}_${name}.json (HIP-event timings)
# usage (on the GPU box, from the repo root): tools/profile_workloads.sh r02 config5 long25 nedm noptrex
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# the HIP-event timings of every workload first: the profiler passes (the --pmc ones in particular) may leave the device in
# another state for whatever runs after them in the same session
for name in "$@"; do
  W=$O/work_$name
  rm -rf $W && mkdir -p $W
  [ -n "$SKIP_PLAIN" ] && continue
  timeout -k 10 300 python3 tools/workload.py $name > $O/${TAG}_${name}.json 2> $W/plain.err
  cat $O/${TAG}_${name}.json
done
for name in "$@"; do
  W=$O/work_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/stats -o s --output-format csv -- python3 tools/workload.py $name --steps 3 > $W/stats.log 2>&1
  python3 profiles/trim_stats.py $W/stats/s_kernel_stats.csv $O/${TAG}_${name}_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $W/pmc_rd -o p --output-format csv -- python3 tools/workload.py $name --steps 2 > $W/pmc_rd.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d $W/pmc_wr -o p --output-format csv -- python3 tools/workload.py $name --steps 2 > $W/pmc_wr.log 2>&1
  DRX_TRAFFIC_SOURCE="rocprofv3 --pmc, tools/workload.py $name, one MI355X" python3 profiles/make_traffic_json.py $O/${TAG}_${name}_pmc_traffic.json $W/pmc_rd/p_counter_collection.csv $W/pmc_wr/p_counter_collection.csv > /dev/null
  head -6 $O/${TAG}_${name}_kernel_stats.csv
  rm -rf $W/stats $W/pmc_rd $W/pmc_wr
done
