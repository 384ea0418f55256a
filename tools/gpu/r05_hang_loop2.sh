#!/bin/bash
# r05, VERDICT r04 item 5: the GPU suite N times back to back with the hang instrumentation of tests/conftest.py (progress file per run, watchdog dump of
# Python stacks + fvsrn_debug_state + child processes after FVSRN_TEST_WATCHDOG_S, thread-method timeouts).  The two CPU-bound test families (the
# reference's 2 160-network matrix, the row bands against the oracle: 180 of the suite's 300 s, no GPU work of their own kind) run in every fifth pass.
N=${1:-20}
O=gpurun_out/r05h2; mkdir -p $O
export FVSRN_TEST_WATCHDOG_S=${FVSRN_TEST_WATCHDOG_S:-150}
for i in $(seq 1 $N); do
  export FVSRN_TEST_PROGRESS=$PWD/$O/progress_$i.log
  t0=$(date +%s)
  if [ $((i % 5)) = 0 ]; then sel=""; else sel="--deselect tests/test_reference_matrix.py -k not(row_band)"; fi
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider $sel > $O/run_$i.txt 2>&1
  rc=$?
  echo "run $i: rc $rc, $(( $(date +%s) - t0 )) s, $(grep -E 'passed|failed|error' $O/run_$i.txt | tail -1)" >> $O/summary.txt
  if [ $rc = 0 ]; then tail -2 $O/run_$i.txt > $O/run_$i.tail; rm -f $O/run_$i.txt $O/progress_$i.log; fi   # (clean runs keep their last lines only)
done
cat $O/summary.txt
