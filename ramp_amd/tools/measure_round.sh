#!/bin/bash
# The round's evidence in one GPU call: the bench line (with roofline + CPU baseline), the rocprofv3 kernel summary of
# the same command, PMC traffic of one evaluation inside a sampling job, and the other configs' bench lines.
# Usage (from the repo root): bash ramp_amd/tools/measure_round.sh [tag]   -> gpurun_out/<tag>_*
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-round}
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench done"
rm -rf gpurun_out/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${T}_prof_bench.json 2> gpurun_out/${T}_prof.err
cp $(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
echo "kernel stats done"
rm -rf gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sample_fetch -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_sample_write -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_w.log 2>&1
python3 ramp_amd/tools/pmc_summary.py gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write gpurun_out/${T}_pmc_traffic.json > gpurun_out/${T}_pmc_summary.log 2>&1
rm -rf gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write
echo "pmc done"
for c in 3 4 5; do python3 bench.py --config $c --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_c$c.json 2> gpurun_out/${T}_bench_c$c.err; done
echo "configs done"
