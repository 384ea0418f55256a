#!/bin/bash
# usage (on the GPU box): tools/pmc_profile.sh <tag> [bench args...]          PMC_PROGRAM="tools/bench_evaluate.py 16777216": profile that program instead of bench.py
# Runs rocprofv3 --kernel-trace --stats once and the PMC passes (separate runs, counters only) for bench.py with the
# given arguments and writes the per-launch summary of the render kernel to gpurun_out/<tag>_{stats,pmc}.csv.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
# (the counter passes render one frame per launch: their per-launch averages are per-frame figures, comparable across rounds; the stats run is the bench's
# own command line -- eight frames per launch since r05 -- and its average duration is roofline.launch_ms_avg of the line)
B="$R/bench.py --steps 6 --warmup 2 --no-twin --no-cpu-baseline --frames-per-submit 1 $*"
S="$R/bench.py --steps 40 --warmup 4 --no-twin --no-cpu-baseline $*"
if [ -n "$PMC_PROGRAM" ]; then B="$R/$PMC_PROGRAM"; S="$R/$PMC_PROGRAM"; fi
# kernel-trace + stats of a run long enough that the warm-up launches (clock ramp) do not dominate the average; the JSON
# line bench.py printed during THIS run is kept next to it (its roofline.kernel_ms_avg is the figure to compare)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $S > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $O/stats.log
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_LEVEL_WAVES" \
  "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCC_HIT TCC_MISS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $B > $O/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $O $tag > $R/gpurun_out/${tag}_pmc.csv
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
cat $R/gpurun_out/${tag}_pmc.csv
