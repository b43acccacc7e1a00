#!/bin/bash
# loud data under the reference's default RiceParameter (m = 8 is too small: escapes, 8-12 bits per sample): tools/r03_loud.sh
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
for w in nab100 noptrex long25 nedm config5; do
for sm in "30 8" "60 8" "120 8" "400 8"; do
  set -- $sm
  timeout -k 10 200 python3 tools/workload.py $w --sigma $1 --m $2 --steps 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w sigma', d['sigma'], 'm', d['m'], 'bits/sample', round(d['ratio'] * 16, 2), 'encode_ms', round(d['encode_ms']['total'], 3), 'frac', round(d['encode_frac_of_8TBps'], 3), 'decode_ms', round(d['decode_ms']['total'], 3), 'frac', round(d['decode_frac_of_8TBps'], 3))"
done
done
