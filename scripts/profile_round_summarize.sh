#!/bin/bash
# Second half of scripts/profile_round.sh, run in the repository (git available) right after the GPU call, from the SAME clean tree:
# writes profiles/rNN_summary.md, rNN_kernel_stats.csv, hbm_traffic.json (with commit + source fingerprint), the config-3 / config-5
# counter tables and copies the bench lines.
set -eu
RND=${1:-r03}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- cmpt-eigenex_amd/csrc cmpt-eigenex_amd/include include bench.py)" ]; then
  echo "sources differ from HEAD: commit first, profile again" >&2; exit 1
fi
export PROFILED_COMMIT=$(git rev-parse --short HEAD)
python3 scripts/summarize_rocprof.py $RND laplacian3d_512_m100_gpus1 gpurun_out/prof_headline_stats gpurun_out/prof_headline_pmc1 gpurun_out/prof_headline_pmc2 > /dev/null
python3 scripts/summarize_pmc.py config3 --out profiles/${RND}_config3_pmc.md --title "BASELINE config 3 (random CSR 1e6 x 32 from std::mt19937_64(12345), split tiles, Arnoldi m = 80): rocprofv3 counters, round ${RND#r}, commit $PROFILED_COMMIT" > /dev/null
python3 scripts/summarize_pmc.py config5 --out profiles/${RND}_config5_pmc.md --title "BASELINE config 5 shape (block Hamiltonian, sectors of 10, N = 2e7 for the counter passes, thick-restart Lanczos m = 128): rocprofv3 counters, round ${RND#r}, commit $PROFILED_COMMIT" > /dev/null
for f in bench bench_128cubed bench_config3 bench_config1 bench_config5 config3_1gpu config5_1gpu; do
  [ -s gpurun_out/${RND}_$f.json ] && cp gpurun_out/${RND}_$f.json profiles/${RND}_$f.json
done
echo "profiles for $RND written from commit $PROFILED_COMMIT"
