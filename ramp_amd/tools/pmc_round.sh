# Per-kernel counters of the timed evaluation: four rocprofv3 --pmc passes over sample_pmc.py, the program directly
# after `--`, no trace domains beside them; merged by pmc_round.py.   usage: bash ramp_amd/tools/pmc_round.sh <tag>  -> gpurun_out/<tag>_pmc_*.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=${1:-r05}
rm -rf gpurun_out/pmc_p1 gpurun_out/pmc_p2 gpurun_out/pmc_p3 gpurun_out/pmc_p4
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_p1 -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_p1.log 2>&1 || echo "pass 1 failed"
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_p2 -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_p2.log 2>&1 || echo "pass 2 failed"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_p3 -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_p3.log 2>&1 || echo "pass 3 failed"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_p4 -- python3 ramp_amd/tools/sample_pmc.py > gpurun_out/${T}_pmc_p4.log 2>&1 || echo "pass 4 failed"
cd ramp_amd/tools && python3 pmc_round.py ../../gpurun_out/${T}_pmc_kernels.json ../../gpurun_out/${T}_pmc_traffic.json ../../gpurun_out/pmc_p1 ../../gpurun_out/pmc_p2 ../../gpurun_out/pmc_p3 ../../gpurun_out/pmc_p4 > ../../gpurun_out/${T}_pmc_summary.txt 2>&1
cd ../.. && rm -rf gpurun_out/pmc_p1 gpurun_out/pmc_p2 gpurun_out/pmc_p3 gpurun_out/pmc_p4
tail -30 gpurun_out/${T}_pmc_summary.txt
