#!/bin/bash
# r05 GPU call 6: the bench line with its new default (8 poses per launch), smoke, the resources of RCCL's kernel at world 1 (for the stand-in of tools/stripe_efficiency.py),
# the SnakeAlt skeletons D / E
O=gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('headline %.2f G  %.4f ms/frame  frac %.4f (min %.4f med %.4f max %.4f)  launch %.3f ms x %d frames' % (d['value']/1e9, d['ms_per_step'], r['frac'], r.get('mfma_frac_min',0), r.get('mfma_frac_median',0), r.get('mfma_frac_max',0), r['launch_ms_avg'], r['frames_per_launch']))
print('single-frame launches %.2f G frac %.4f' % (d['single_frame_launches']['value']/1e9, d['single_frame_launches']['mfma_frac']))
print('twin %.2f G  exact %.2f G  issue %s  traffic_ratio %s  cpu %s' % (d['twin']['value']/1e9, d['exact_features']['value']/1e9, (r.get('issue') or {}).get('frac'), r.get('traffic_ratio'), d.get('cpu_baseline',{}).get('value')))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | cut -c1-300 > $O/smoke.txt; cat $O/smoke.txt | cut -c1-200
rocprofv3 --kernel-trace --output-format csv -d $O/trace_fc -- python3 bench.py --force-collective --steps 16 --warmup 8 --no-twin --no-cpu-baseline --frames-per-submit 1 > $O/bench_fc_k1.json 2> $O/fc.err
python - <<'PY' > gpurun_out/r05e/rccl_kernel_resources.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r05e/trace_fc/*/*kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "nccl" in k.lower() or "rccl" in k.lower():
        acc[(k[:80], row.get("Grid_Size_X", row.get("Grid_Size")), row.get("Workgroup_Size_X", row.get("Workgroup_Size")), row.get("LDS_Block_Size"), row.get("VGPR_Count"), row.get("SGPR_Count"))].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
for k, v in acc.items():
    print("kernel %s grid %s workgroup %s lds %s vgpr %s sgpr %s: %d launches, %.1f us avg (world 1: a device-local copy of 32 MiB per frame)" % (k + (len(v), sum(v) / len(v))))
PY
cat $O/rccl_kernel_resources.txt; rm -rf $O/trace_fc
./tools/microbench/r03_snakealt > $O/snakealt_microbench_r05.txt 2>&1; cat $O/snakealt_microbench_r05.txt
