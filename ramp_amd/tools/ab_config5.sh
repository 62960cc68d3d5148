set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for L in old.so new.so; do
    RAMP_HIP_LIB=$GRAFT_REPO_ROOT/ramp_amd/lib/alt/$L timeout -k 10 300 python bench.py --config 5 --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), d.get('range_flag'))"
  done
done
