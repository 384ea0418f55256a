#!/bin/bash
# same-box A/B of the r02 tree (commit f586d69, built here) against the r03 tree: headline, configs[1], configs[2], configs[3]
O=gpurun_out/r03c12; mkdir -p $O
for i in 1 2 3; do
  for c in c32l4_fourier_1024x512 c32l4_fourier_512x256 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do
    (cd gpurun_ab_r02 && bash tools/quick_bench.sh r02 --config $c --no-twin)
    bash tools/quick_bench.sh r03 --config $c --no-twin
  done
done 2>&1 | tee $O/ab_r02_r03.txt
