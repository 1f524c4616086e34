set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synthetic.py -x -q -m gpu 2>&1 | tail -3
for g in 0 1; do
  echo "FUSE_F=$g"
  HIFIR_AMD_FUSE_F=$g timeout -k 10 200 python tests/perf_probe.py 1000 default 64 10 2>&1 | grep -E "RESULT|relerr|launches" 
done
HIFIR_AMD_FUSE_F=1 timeout -k 10 200 python tests/perf_probe.py 1000 tuned 64 10 2>&1 | grep -E "RESULT|relerr|launches"
