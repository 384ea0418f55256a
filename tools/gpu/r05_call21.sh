#!/bin/bash
# r05 GPU call 21: same-box A/B of the committed binary (with the two A/B build switches in kernels.hpp) and the cleaned one (switches removed): headline, twin, evaluate_points
O=gpurun_out/r05t; mkdir -p $O
for round in 1 2 3; do for v in committed cleaned; do
  L=$PWD/fv-srn_amd/ablate/libfvsrn_committed.so; [ $v = cleaned ] && L=$PWD/fv-srn_amd/libfvsrn.so
  FVSRN_LIBRARY=$L python bench.py --no-cpu-baseline 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v headline %.2f G frac %.4f single %.2f twin %.2f exact %.2f' % (d['value']/1e9, d['roofline']['frac'], d['single_frame_launches']['value']/1e9, d['twin']['value']/1e9, d['exact_features']['value']/1e9))" | tee -a $O/ab.txt
  FVSRN_LIBRARY=$L python tools/bench_evaluate.py 16777216 c32l4_fourier_relu c32l4_fourier_snakealt 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin: d=json.loads(l); print('$v', d['workload'], '%.1f G points/s' % (d['points_per_s']/1e9))" | tee -a $O/ab.txt
done; done
