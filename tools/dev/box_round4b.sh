#!/bin/bash
# Box script: cell-table A/B (suite + benches), the op_sel sweep incl. v_pk_mov_b32, the RCCL one-rank line (stdout must hold ONE line), smoke()
mkdir -p gpurun_out
bash tools/dev/cells_ab.sh
timeout 300 tools/microbench/bin/r04_pk_opsel_sweep > gpurun_out/repro3b_op_sel_sweep_with_pk_mov.txt 2>&1; echo "sweep rc $?"
grep -c . gpurun_out/repro3b_op_sel_sweep_with_pk_mov.txt; grep -i "mov" gpurun_out/repro3b_op_sel_sweep_with_pk_mov.txt | head -20
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --force-collective --no-twin --no-cpu-baseline > gpurun_out/force_collective_stdout.txt 2> gpurun_out/force_collective_stderr.txt; echo "force-collective rc $? stdout lines: $(wc -l < gpurun_out/force_collective_stdout.txt)"
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
