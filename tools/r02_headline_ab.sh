#!/bin/bash
# headline decode/encode kernel times for library variants, interleaved: tools/r02_headline_ab.sh variant...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
for rep in 1 2 3; do for v in "$@"; do
  echo "== $v" >> $O/ab.txt
  DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$v.so timeout -k 10 120 python3 bench.py --cpu-seconds 0 --steps 10 --warmup 3 $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'dec_ms': d['kernel_ms']['decode_kernel'], 'enc_ms': d['kernel_ms']['encode_kernel'], 'frac': d['roofline']['frac']}))" >> $O/ab.txt
done; done
python3 - <<'PY'
import json,os,collections
R=os.environ.get("GRAFT_REPO_ROOT",os.getcwd()); cur=None; res=collections.defaultdict(list)
for ln in open(f"{R}/gpurun_out/r02h/ab.txt"):
    if ln.startswith("== "): cur=ln.split()[1]
    elif ln.startswith("{"): d=json.loads(ln); res[cur].append(d)
for k,v in res.items(): print(k, "decode ms:", " ".join(f"{d['dec_ms']:.3f}" for d in v), "| encode ms:", " ".join(f"{d['enc_ms']:.3f}" for d in v))
PY
