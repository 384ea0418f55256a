#!/bin/bash
# Box script: the two-rank gloo run of the time-dependent latent-grid configuration (stripes + cell tables + two working grids), then the round profile
mkdir -p gpurun_out
FVSRN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --config c64l6_grid16_time16_1024x512 --steps 3 --warmup 1 --spinup-ms 0 --no-twin --no-cpu-baseline > gpurun_out/two_rank_gloo_c64l6_time16_r04.json 2> gpurun_out/two_rank_gloo_c64l6_time16_r04.err
python -c "
import json
d=json.loads(open('gpurun_out/two_rank_gloo_c64l6_time16_r04.json').read().strip().splitlines()[-1])
print('two-rank gloo c64l6 time16: frames match', d['gathered_frame_matches_single_gpu_frame'], d['kernel'], d['value']/1e9)"
bash tools/run_round_profile.sh r04 > gpurun_out/round_profile_r04.log 2>&1; tail -14 gpurun_out/round_profile_r04.log
