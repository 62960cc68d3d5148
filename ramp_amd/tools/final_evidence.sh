# Round-end evidence on one GPU box: the GPU suite, rocprofv3 kernel statistics, the two HBM-traffic counter passes, the default bench line.
# (run as: gpurun --timeout 1200 -- bash ramp_amd/tools/final_evidence.sh; results under gpurun_out/fe_*)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/fe_gpu_tests.log 2>&1
echo tests-done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline > gpurun_out/fe_stats_bench.log 2>&1
echo stats-done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fe_pmc_fetch -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/fe_pmc_f.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fe_pmc_write -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/fe_pmc_w.log 2>&1
python3 ramp_amd/tools/pmc_summary.py gpurun_out/fe_pmc_fetch gpurun_out/fe_pmc_write gpurun_out/fe_pmc_traffic.json > gpurun_out/fe_pmc_sum.log 2>&1
rm -rf gpurun_out/fe_pmc_fetch gpurun_out/fe_pmc_write
find gpurun_out/fe_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/fe_kernel_stats.csv \;
rm -rf gpurun_out/fe_stats
echo pmc-done
timeout -k 10 400 python bench.py > gpurun_out/fe_bench.json 2> gpurun_out/fe_bench.err
tail -c 600 gpurun_out/fe_bench.json
