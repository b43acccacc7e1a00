#!/bin/bash
# general filters behind the block decoder: fused inverse filter (default) against the two-pass form (flag 2097152) and the delta counterparts
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_fir4; mkdir -p $O; cd $R; : > $O/table.txt
for rep in 1 2; do
for w in noptrex noptrex_fir4 nedm nedm_fir4 raglong raglong_fir4; do
  for flags in 0 2097152; do
    case $w in *fir4) ;; *) [ $flags = 2097152 ] && continue;; esac
    timeout -k 10 200 python3 tools/workload.py $w --steps 5 --debug-flags $flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-14s flags %-8s decode %.3f ms  encode %.3f ms' % ('$w', '$flags', d['decode_ms']['total'], d['encode_ms']['total']))" | tee -a $O/table.txt
  done
done; done
