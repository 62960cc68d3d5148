# Same-box A/B of launch-plan defaults through the environment (read once in ramp_create): alternates `env A` / `env B` runs of the default job.
# usage (on the GPU box): bash ramp_amd/tools/ab_env.sh "RAMP_TKL=65536" "RAMP_TKL=40000" [rounds]
cd "${GRAFT_REPO_ROOT:-.}"
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E timeout -k 10 240 python bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', round(d['value'],1))"
  done
done
