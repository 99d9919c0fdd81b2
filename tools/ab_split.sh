set -e
for rep in 1 2 3; do
for v in 0 1; do
  ODEF_SMOOTH_SPLIT=$v python tools/bench_modes.py --traj ${1:-2048} --nsteps 64 --modes pleiades_smooth 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v', round(d['smooth_ms'],1))
"
done; done
