#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/adv_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/adv_pytest.log
tail -4 gpurun_out/adv_pytest.log
for i in 1 2; do tools/quick_bench.sh main --steps 40 --warmup 4 --no-twin; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('main with twins: %.2f | twin %.2f | exact %.2f (%.3f)' % (d['value']/1e9, d['twin']['value']/1e9, d['exact_features']['value']/1e9, d['exact_features']['mfma_frac']))"
python bench.py --config c32l4_grid16_1024x512 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grid main: %.2f | twin %.2f | exact %.2f (%.3f)' % (d['value']/1e9, d['twin']['value']/1e9, d['exact_features']['value']/1e9, d['exact_features']['mfma_frac']))"
