#!/bin/bash
# k_encode_stream (encode_impl 2) against k_encode_fused (1) off the headline: WaveformLengths of the single pass's range, the
# AR(1) workloads of BASELINE config 3, noisy data with its RiceParameter, one chunk.  usage: tools/r04_enc_regress.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_enc_regress; mkdir -p $O; cd $R; : > $O/table.txt
for impl in 2 1 2 1; do
  echo "== encode_impl $impl: len_sweep, 100 chunks of 14 M samples" | tee -a $O/table.txt
  DRX_ENCODE_IMPL=$impl DRX_SWEEP_CHUNKS=100 timeout -k 10 300 python3 tools/len_sweep.py 3600 4096 5000 7000 8192 10000 2>&1 | tail -7 | tee -a $O/table.txt
done
for impl in 2 1; do
  echo "== encode_impl $impl: bench.py --dist ar1, m = 4 / 8 / 16" | tee -a $O/table.txt
  for m in 4 8 16; do
    DRX_ENCODE_IMPL=$impl timeout -k 10 300 python3 bench.py --no-collect --cpu-seconds 0 --steps 5 --warmup 2 --dist ar1 --m $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('m=$m  enc %.3f ms  dec %.3f ms  ratio %.4f' % (d['kernel_ms']['encode_kernel'], d['kernel_ms']['decode_kernel'], d['compression_ratio']))" | tee -a $O/table.txt
  done
  echo "== encode_impl $impl: small batches (waveforms per GPU)" | tee -a $O/table.txt
  for w in 2000 20000 100000; do
    DRX_ENCODE_IMPL=$impl timeout -k 10 300 python3 bench.py --no-collect --cpu-seconds 0 --steps 20 --warmup 5 --waves $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('waves=$w  enc %.4f ms  dec %.4f ms' % (d['kernel_ms']['encode_kernel'], d['kernel_ms']['decode_kernel']))" | tee -a $O/table.txt
  done
done
