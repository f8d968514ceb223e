#!/bin/bash
# The round's tracked profiles, mechanically, from the tree that is on the GPU box.  Run as the LAST GPU call of a round, from a
# clean tree:   gpurun --timeout 1150 -- scripts/profile_round.sh r03
# then, here (where git is):   scripts/profile_round_summarize.sh r03
# Passes (each its own rocprofv3 run; a killed pass stops the script: no GPU step after a timeout):
#   headline  bench.py 512^3 m=100: kernel trace, FETCH_SIZE, WRITE_SIZE          -> profiles/rNN_summary.md, hbm_traffic.json
#   config3   scripts/run_config3.py: + L2 (TCC) and L1 (TCP) request counters    -> profiles/rNN_config3_pmc.md
#   config5   scripts/run_config5.py at N = 2e7: + L2 counters                    -> profiles/rNN_config5_pmc.md
# and plain (un-profiled) bench lines for the record: headline, 128^3, config 3, config 1, config 5, config 3 / 5 script JSONs.
set -u
RND=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
L2="TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum"
L1="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
scripts/profile_passes.sh headline "FETCH_SIZE;WRITE_SIZE" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline || exit $?
scripts/profile_passes.sh config3 "FETCH_SIZE;WRITE_SIZE;$L2;$L1" python3 scripts/run_config3.py 1000000 80 2 || exit $?
PASS_TIMEOUT=600 scripts/profile_passes.sh config5 "FETCH_SIZE;WRITE_SIZE;$L2" python3 scripts/run_config5.py 20000000 10 4 6 || exit $?
run() { echo "== $*" >&2; timeout -k 10 500 "$@" || exit $?; }
run python3 bench.py --steps 5 --warmup 2 > gpurun_out/${RND}_bench.json 2> gpurun_out/${RND}_bench.err
run python3 bench.py --grid-edge 128 --krylov-steps 50 --steps 20 --warmup 5 > gpurun_out/${RND}_bench_128cubed.json 2> gpurun_out/${RND}_bench_128cubed.err
run python3 bench.py --workload config3 --steps 5 --warmup 2 > gpurun_out/${RND}_bench_config3.json 2> gpurun_out/${RND}_bench_config3.err
run python3 bench.py --workload config1 --steps 5 --warmup 2 > gpurun_out/${RND}_bench_config1.json 2> gpurun_out/${RND}_bench_config1.err
run python3 bench.py --workload config5 --steps 2 --warmup 1 > gpurun_out/${RND}_bench_config5.json 2> gpurun_out/${RND}_bench_config5.err
run python3 scripts/run_config3.py 1000000 80 3 --no-profile --json gpurun_out/${RND}_config3_1gpu.json > gpurun_out/${RND}_config3.log 2>&1
run python3 scripts/run_config5.py --json gpurun_out/${RND}_config5_1gpu.json > gpurun_out/${RND}_config5.log 2>&1
tail -c 400 gpurun_out/${RND}_bench.json
