# A/B of the block decoder: the parity tests that reach it, then decode times of the three long-waveform workloads with
# deltarice_amd/variants/lib_base.so (the tree before the change) and the tree's library, interleaved.  usage: r04_blk_ab.sh [steps]
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_long_filters.py tests/test_gpu_noise_levels.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for w in ${WORKLOADS:-long25 nedm noptrex noptrex_fir4 nedm_fir4}; do
  for lib in base new base new; do
    if [ $lib = base ]; then export DRX_LIB_PATH=$PWD/deltarice_amd/variants/lib_base.so; else unset DRX_LIB_PATH; fi
    timeout -k 10 200 python3 tools/workload.py $w --steps ${1:-8} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $lib', 'encode', round(d['encode_ms']['total'],3), 'decode', round(d['decode_ms']['total'],3))"
  done
done
