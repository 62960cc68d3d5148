cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06b_stats5 -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r06b_stats5_bench.log 2>&1
find gpurun_out/r06b_stats5 -name '*kernel_stats.csv' -exec cp {} gpurun_out/r06b_config5_kernel_stats.csv \;
rm -rf gpurun_out/r06b_stats5
echo stats5-done
timeout -k 10 500 python bench.py > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
echo bench-done
tail -c 300 gpurun_out/r06b_bench.json
