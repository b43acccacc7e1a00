#!/bin/bash
# headline decode/encode kernel times for library variants, interleaved: tools/r03_headline_ab.sh TAG variant...
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift; O=$R/gpurun_out/r03_ab_$TAG; mkdir -p $O; cd $R; : > $O/ab.txt
for rep in 1 2 3; do for v in "$@"; do
  echo "== $v" >> $O/ab.txt
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$v.so timeout -k 10 120 python3 bench.py --no-collect --cpu-seconds 0 --steps 10 --warmup 3 $EXTRA 2>$O/err_$v.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'dec_ms': d['kernel_ms']['decode_kernel'], 'enc_ms': d['kernel_ms']['encode_kernel'], 'frac': d['roofline']['frac']}))" >> $O/ab.txt
done; done
python3 - $O/ab.txt <<'PY'
import json,sys,collections
cur=None; res=collections.defaultdict(list)
for ln in open(sys.argv[1]):
    if ln.startswith("== "): cur=ln.split()[1]
    elif ln.startswith("{"): res[cur].append(json.loads(ln))
for k,v in res.items(): print("%-14s decode ms: %s | encode ms: %s" % (k, " ".join(f"{d['dec_ms']:.3f}" for d in v), " ".join(f"{d['enc_ms']:.3f}" for d in v)))
PY
