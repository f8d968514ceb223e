#!/bin/bash
# The round's tracked profiles, mechanically, from the tree that is on the GPU box (run as the LAST GPU call of a round, from a
# clean tree: `gpurun --timeout 1100 -- scripts/profile_round.sh r03`, then `PROFILED_COMMIT=$(git rev-parse --short HEAD)
# python scripts/summarize_rocprof.py r03 laplacian3d_512_m100_gpus1 gpurun_out/prof_headline_stats gpurun_out/prof_headline_pmc1
# gpurun_out/prof_headline_pmc2` here, where git is).  Passes: headline (kernel trace + FETCH_SIZE + WRITE_SIZE), then a plain
# bench line for the record.  Each pass is its own run; a killed pass stops the script (no GPU step after a timeout).
set -u
RND=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
scripts/profile_passes.sh headline "FETCH_SIZE;WRITE_SIZE" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline || exit $?
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > gpurun_out/${RND}_bench.json 2> gpurun_out/${RND}_bench.err || exit $?
tail -c 600 gpurun_out/${RND}_bench.json
