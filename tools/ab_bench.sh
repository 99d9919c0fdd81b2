set -e
for rep in 1 2 3; do
for v in base new; do
  if [ $v = base ]; then export ODEFILTER_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/base.so; else unset ODEFILTER_HIP_LIB; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['parity_ok'])
"
done; done
