#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/suite_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/suite_pytest.log
tail -5 gpurun_out/suite_pytest.log
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline: %.2f (%.3f) | twin %.2f | exact %.2f (%.3f)' % (d['value']/1e9, d['roofline']['frac'], d['twin']['value']/1e9, d['exact_features']['value']/1e9, d['exact_features']['mfma_frac']))"
