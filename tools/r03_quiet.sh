#!/bin/bash
# quiet data under a RiceParameter chosen for louder data (many short codes per lane): tools/r03_quiet.sh
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
for w in noptrex long25 nab100 config5; do
for sm in "0 8" "1 8" "3 8" "10 8" "10 64" "40 1024"; do
  set -- $sm
  timeout -k 10 200 python3 tools/workload.py $w --sigma $1 --m $2 --steps 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w sigma', d['sigma'], 'm', d['m'], 'bits/sample', round(d['ratio'] * 16, 2), 'encode_ms', round(d['encode_ms']['total'], 3), 'frac', round(d['encode_frac_of_8TBps'], 3), 'decode_ms', round(d['decode_ms']['total'], 3), 'frac', round(d['decode_frac_of_8TBps'], 3))"
done
done
