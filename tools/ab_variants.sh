set -e
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export ODEFILTER_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/base.so; else unset ODEFILTER_HIP_LIB; fi
  python tools/configs_r02.py --only 2,3s,5,3 --reps 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d['config'][:30], round(d.get('filter_ms',0),3), round(d.get('smooth_ms',0),3))
"
done; done
