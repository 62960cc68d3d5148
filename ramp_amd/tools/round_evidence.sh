# Round evidence on one GPU box: rocprofv3 kernel statistics of the bench command (config 2) and of config 5's shard, the per-kernel counter
# passes (pmc_round.sh), the default bench line.   usage: gpurun --timeout 1200 -- bash ramp_amd/tools/round_evidence.sh r05  -> gpurun_out/r05_*
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-r05}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline > gpurun_out/${T}_stats_bench.log 2>&1
find gpurun_out/${T}_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_kernel_stats.csv \;
rm -rf gpurun_out/${T}_stats
echo stats-done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats5 -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${T}_stats5_bench.log 2>&1
find gpurun_out/${T}_stats5 -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_config5_kernel_stats.csv \;
rm -rf gpurun_out/${T}_stats5
echo stats5-done
bash ramp_amd/tools/pmc_round.sh ${T} > gpurun_out/${T}_pmc_tail.txt 2>&1
echo pmc-done
timeout -k 10 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo bench-done
tail -c 400 gpurun_out/${T}_bench.json
