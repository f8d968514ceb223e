#!/bin/bash
# rocprofv3 passes over one workload command, each in its own run (counters never share a run with the trace):
#   scripts/profile_passes.sh NAME "COUNTERS_1;COUNTERS_2;..." PROGRAM ARGS...
# writes gpurun_out/prof_NAME_stats (--kernel-trace --stats) and gpurun_out/prof_NAME_pmcK (one per counter group).
# PROGRAM must be the program itself (python3 ...): no env/bash -c hop between rocprofv3 and it.
# A pass that fails (e.g. counters that do not fit one pass) does not stop the others; a pass that is KILLED
# (timeout) does.
set -u
NAME=$1; GROUPS_=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
run() {  # dir, rocprof args...
  local d=$1; shift
  rm -rf "$OUT/$d"
  timeout -k 10 ${PASS_TIMEOUT:-420} rocprofv3 "$@" --output-format csv -d "$OUT/$d" -- "${CMD[@]}" > "$OUT/$d.log" 2>&1
  local rc=$?
  echo "pass $d rc=$rc" | tee -a $OUT/prof_${NAME}_passes.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $d was killed: stopping"; exit $rc; fi
  # keep only the small per-kernel summaries from the trace pass
  find "$OUT/$d" -name '*_kernel_trace.csv' -size +8M -delete 2>/dev/null
  return 0
}
CMD=("$@")
: > $OUT/prof_${NAME}_passes.txt
run prof_${NAME}_stats --kernel-trace --stats
k=0
IFS=';' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  [ -z "$g" ] && continue
  k=$((k+1))
  # shellcheck disable=SC2086
  run prof_${NAME}_pmc$k --pmc $g
done
