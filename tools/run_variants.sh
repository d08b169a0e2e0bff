# Tests and benches each library under arbitrage-free-iv-surfaces-using-vae_amd/variants/ (GPU box).
set -e
mkdir -p gpurun_out
for lib in arbitrage-free-iv-surfaces-using-vae_amd/variants/*.so; do
  v=$(basename $lib .so)
  export AFIV_LIB_PATH=$PWD/$lib
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_math.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -1
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$v.json
  python -c "
import json;d=json.load(open('gpurun_out/bench_$v.json'));print('$v',int(d['value']),d['roofline']['kernel_ms'],round(d['roofline']['frac'],3),d.get('roofline_iv_checks',{}).get('kernel_ms'))"
done
