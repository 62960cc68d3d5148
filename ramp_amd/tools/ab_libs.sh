# Same-box A/B of two builds of libramp_hip.so: runs the default job with RAMP_HIP_LIB pointing at ramp_amd/lib/alt/<name>.so (the in-tree
# library is never overwritten), twice per library, alternating.
# usage (on the GPU box): bash ramp_amd/tools/ab_libs.sh libA.so libB.so [rounds]
set -e
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for L in $A $B; do
    RAMP_HIP_LIB=$GRAFT_REPO_ROOT/ramp_amd/lib/alt/$L timeout -k 10 240 python bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1))"
  done
done
