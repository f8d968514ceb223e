#!/bin/bash
# rocprofv3 passes over the headline workload (bench.py, 512^3 m=100): kernel trace + FETCH_SIZE + WRITE_SIZE, each its own run
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
scripts/profile_passes.sh headline "FETCH_SIZE;WRITE_SIZE" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
