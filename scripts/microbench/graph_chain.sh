#!/bin/bash
# one chain length per process, ascending, stop at the first that fails; then the failing length once more with an
# unlimited stack (if the failure is a stack overflow of a recursive walk over the chain, that run passes)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/graph_chain.log
: > $OUT
for n in 2000 20000 60000 120000 180000 260000; do
  echo "=== nodes $n (default stack)" >> $OUT
  timeout -k 5 120 scripts/microbench/graph_chain $n >> $OUT 2>&1
  rc=$?
  echo "rc=$rc" >> $OUT
  if [ $rc -ne 0 ]; then
    echo "=== nodes $n (ulimit -s unlimited)" >> $OUT
    ( ulimit -s unlimited; timeout -k 5 120 scripts/microbench/graph_chain $n ) >> $OUT 2>&1
    echo "rc=$?" >> $OUT
    echo "=== nodes $n (ulimit -s 262144)" >> $OUT
    ( ulimit -s 262144; timeout -k 5 120 scripts/microbench/graph_chain $n ) >> $OUT 2>&1
    echo "rc=$?" >> $OUT
    break
  fi
done
grep -E "===|rc=|SIGNAL|stage|failed" $OUT | tail -60
