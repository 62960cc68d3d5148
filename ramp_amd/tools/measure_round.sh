set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
true

rm -rf gpurun_out/r2b_prof
rm -rf gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sample_fetch -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/r2b_pmc_f.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_sample_write -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/r2b_pmc_w.log 2>&1
echo "write done"
python3 ramp_amd/tools/pmc_summary.py gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write gpurun_out/r2b_pmc_traffic.json > gpurun_out/r2b_pmc_summary.log 2>&1
rm -rf gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write gpurun_out/r2b_prof
for c in 3 4 5; do python3 bench.py --config $c --no-cpu-baseline --no-roofline > gpurun_out/r2b_bench_c$c.json 2> gpurun_out/r2b_bench_c$c.err; tail -c 300 gpurun_out/r2b_bench_c$c.json; done
