#!/bin/bash
# A/B/... of several builds of libafiv.so on ONE GPU box (boxes differ by about 1 %, more than most
# kernel changes): put the libraries in ab_tmp/ as libafiv_<name>.so (built here; *.so files travel
# with the snapshot), then   gpurun -- 'bash tools/ab_compare.sh'
# Prints, per run: variant, ms per step, pricing kernel ms, implied-vol kernel ms; three rounds over
# all variants.  Restore the library of the tree afterwards (make -C .../csrc).
L=arbitrage-free-iv-surfaces-using-vae_amd/libafiv.so
cp $L ab_tmp/.saved.so
for round in 1 2 3; do
  for f in ab_tmp/libafiv_*.so; do
    v=${f#ab_tmp/libafiv_}; v=${v%.so}
    cp $f $L
    timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={d['roofline']['kernel']: d['roofline']['kernel_ms_per_step'], d['roofline_other_kernel']['kernel']: d['roofline_other_kernel']['kernel_ms_per_step']}
print('$v', round(d['ms_per_step'],3), round(k['price_kernel'],3), round(k['iv_fast_kernel'],3))" || exit 1
  done
done
cp ab_tmp/.saved.so $L
