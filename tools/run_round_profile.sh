#!/bin/bash
# usage (GPU box): tools/run_round_profile.sh <round tag>      everything the round's profiles/ directory is built from -> gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-r05}
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > gpurun_out/smoke_${V}.txt
grep "2-rank line" gpurun_out/smoke_${V}.txt | sed 's/^smoke 2-rank line: //' > gpurun_out/two_rank_gloo_${V}.json
# the PMC profiles first: the bench lines below take roofline.traffic / roofline.issue of a workload from the newest profiles/r*/<workload>_rNN_pmc.csv
mkdir -p profiles/${V}
for c in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/pmc_profile.sh ${c}_${V} --config $c > /dev/null 2>&1; done
tools/pmc_profile.sh c32l4_fourier_snakealt_1024x512_${V} --config c32l4_fourier_1024x512 --activation SnakeAlt > /dev/null 2>&1
cp gpurun_out/*_${V}_pmc.csv profiles/${V}/ 2>/dev/null
python bench.py > gpurun_out/bench_${V}.json 2> gpurun_out/bench_${V}.err
python bench.py --activation SnakeAlt --no-cpu-baseline > gpurun_out/bench_${V}_snakealt.json 2>/dev/null
# one launch per frame (what r01 - r04 timed): the frame-by-frame figure of the headline with its own roofline object
python bench.py --frames-per-submit 1 --no-cpu-baseline > gpurun_out/bench_${V}_single_frame_launches.json 2>/dev/null
# the RCCL route on one GPU (one-rank process group, compact stripes + all_gather_into_tensor + assemble; per-rank render / gather times)
python bench.py --force-collective --no-twin --no-cpu-baseline > gpurun_out/bench_${V}_force_collective_nccl.json 2>/dev/null
python bench.py --force-collective --no-twin --no-cpu-baseline --config c64l6_grid16_time16_1024x512 > gpurun_out/bench_${V}_force_collective_nccl_c64l6_time16.json 2>/dev/null
python bench.py --force-collective --no-twin --no-cpu-baseline --gather root --payload rgba8 > gpurun_out/bench_${V}_force_collective_nccl_root_rgba8.json 2>/dev/null
for c in c32l4_fourier_512x256 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${V}_$c.json 2>/dev/null; done
# the latent-grid lines on the gather path (FVSRN_OPT_CELL_TABLE = 0: what r01 - r03 ran), for the A/B of the cell table
for c in c32l4_grid16_1024x512 c64l6_grid16_1024x512; do FVSRN_CELL_TABLE=0 python bench.py --config $c --no-cpu-baseline --no-twin > gpurun_out/bench_${V}_${c}_gather_path.json 2>/dev/null; done
python - <<'PY'
import sys
sys.path.insert(0, '.')
import fvsrn_amd
from fvsrn_amd import synthetic, volnet_io
for act in ("ReLU", "SnakeAlt"):
    vn = synthetic.random_network(C=32, layers=4, activation=act, param=1.0, output_mode="density:direct", seed=1234, box_min=(-0.5, -0.5, -0.5))
    open("/tmp/protocol_%s.volnet" % act, "wb").write(volnet_io.save_volnet(vn))
PY
for a in ReLU SnakeAlt; do python tools/render_protocol.py /tmp/protocol_$a.volnet 2>/dev/null | tail -1 >> gpurun_out/protocol_c32l4_512x512_${V}.jsonl; done
# evaluate_points: throughput with both rooflines, and the rocprofv3 kernel summary of the same command
PMC_PROGRAM="tools/bench_evaluate.py 16777216 c32l4_fourier_relu" tools/pmc_profile.sh evaluate_points_c32l4_fourier_relu_${V} > /dev/null 2>&1
cp gpurun_out/evaluate_points_c32l4_fourier_relu_${V}_pmc.csv profiles/ 2>/dev/null  # (bench_evaluate.py reads the PMC summary of its kernel from profiles/r*/: a flat copy for this call)
mkdir -p profiles/${V}; cp gpurun_out/evaluate_points_c32l4_fourier_relu_${V}_pmc.csv profiles/${V}/ 2>/dev/null
python tools/bench_evaluate.py > /dev/null 2>&1; python tools/bench_evaluate.py > gpurun_out/bench_evaluate_${V}.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_evaluate_${V} -- python3 tools/bench_evaluate.py 16777216 > gpurun_out/bench_evaluate_under_rocprof_${V}.jsonl 2>/dev/null
cp $(ls gpurun_out/prof_evaluate_${V}/*/*kernel_stats.csv | head -1) gpurun_out/evaluate_points_${V}_kernel_stats.csv 2>/dev/null
python tools/bench_evaluate_gradients.py > gpurun_out/bench_evaluate_gradients_${V}.jsonl 2>/dev/null
python tools/bench_shaded.py > gpurun_out/bench_shaded_${V}.jsonl 2>/dev/null
FVSRN_STRIPE_BATCH=8 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512 > gpurun_out/stripe_efficiency_${V}.jsonl 2>/dev/null
FVSRN_STRIPE_BATCH=1 python tools/stripe_efficiency.py > gpurun_out/stripe_efficiency_frame_by_frame_${V}.jsonl 2>/dev/null
FVSRN_WORKING_GRIDS=1 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 > gpurun_out/stripe_efficiency_one_working_grid_${V}.jsonl 2>/dev/null
# the same with a stand-in for the collective on the comm stream (tools/dev/occupy.hip), with ROCm's default of four hardware queues, and with the r02 launch shape
if [ -f tools/dev/bin/liboccupy.so ]; then
  FVSRN_STRIPE_BATCH=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,100 python tools/stripe_efficiency.py > gpurun_out/stripe_efficiency_with_stand_in_${V}.jsonl 2>/dev/null
  python tools/dev/coschedule.py 2>/dev/null > gpurun_out/coschedule_${V}.txt
fi
python tools/bench_grid_volume.py > gpurun_out/grid_volume_bench_${V}.json 2>/dev/null
python tools/bench_tail_variants.py 2>/dev/null | grep -v amdgpu > gpurun_out/tail_variants_${V}.txt
python tools/dev/stress_concurrent.py 100 2>&1 | grep -v amdgpu.ids > gpurun_out/stress_concurrent_${V}.txt
python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids > gpurun_out/determinism_${V}.txt
python tests/test_fuzz_parity.py 400 > gpurun_out/fuzz_report_gpu_vs_device_model_${V}.txt 2>/dev/null
# kernels off the happy path (bench variants): BYTE_GAUSSIAN grids, the adjoint / finite-difference shaded renders of the 64-wide network
for c in c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do python bench.py --config $c --grid-encoding byte_gaussian --no-cpu-baseline > gpurun_out/bench_${V}_${c}_byte_gaussian.json 2>/dev/null; done
for m in finite_differences adjoint; do python bench.py --config c64l6_grid16_1024x512 --gradient-mode $m --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${V}_c64l6_grid16_1024x512_$m.json 2>/dev/null; done
tools/pmc_profile.sh c64l6_grid16_1024x512_byte_gaussian_${V} --config c64l6_grid16_1024x512 --grid-encoding byte_gaussian > /dev/null 2>&1
tools/pmc_profile.sh c64l6_grid16_1024x512_adjoint_${V} --config c64l6_grid16_1024x512 --gradient-mode adjoint > /dev/null 2>&1
for f in gpurun_out/bench_${V}*.json; do python -c "
import json,sys
d=json.load(open('$f'))
print(d['config']['workload'].split(':')[0], '%.2f G %.3f ms frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac']), 'twin %.2f' % (d['twin']['value']/1e9) if d.get('twin') else '', 'exact %.2f' % (d['exact_features']['value']/1e9) if d.get('exact_features') else '', d.get('cpu_baseline',{}).get('value'))"; done
rm -rf gpurun_out/prof_*   # (the summaries above are what is kept; the raw traces would overflow the 64 MiB that travel back)
