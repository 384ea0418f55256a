#!/bin/bash
# r05 GPU call 20: the headline with exact features re-derived every 64 (default) / 128 / 256 / 512 steps (FVSRN_FOURIER_RESYNC), interleaved, two rounds
O=gpurun_out/r05r; mkdir -p $O
for round in 1 2; do for k in 64 128 256 512; do
  FVSRN_FOURIER_RESYNC=$k python bench.py --no-cpu-baseline --no-twin 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('resync $k: %.2f G frac %.4f  launch info %s' % (d['value']/1e9, d['roofline']['frac'], d['launch']))" | tee -a $O/resync.txt
done; done
