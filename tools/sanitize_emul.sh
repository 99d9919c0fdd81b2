#!/bin/bash
# CPU-only: builds the host emulation of the device source (tests/emul) with AddressSanitizer + UBSan and drives the
# lane filter (fixed + adaptive), the lane and row smoothers, dense output, the sampler and the tiled Pleiades filter
# through it.  (GPU sanitizers are not available on the pool; the per-lane source is the same.)
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++20 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-unknown-pragmas \
    tests/emul/emul.cpp -o /tmp/libodef_emul_san.so
cat > /tmp/odef_san_run.py <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import _emul as E
E._LIB = C.CDLL('/tmp/libodef_emul_san.so')
from oracle import odefilter_oracle as orc
vf = orc.vector_field("lorenz63")
tg = np.arange(17) * 2.0**-7
ds = (np.linspace(0.0, tg[-1], 23), 2, 5, 1.0)  # dense-grid sampling
E.emul_solve(vf.rhs_id, 3, 3, True, orc.ensemble_u0(vf.u0, 3, 1e-2), vf.p, tgrid=tg, smooth=True, sample=(2, 5, 1.0), dense_t=[0.01, 0.05],
             dense_sample=ds)
E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, adaptive=True, t0=0.0, t1=0.25, dt0=2.0**-9, max_save=128, smooth=True,
             sample=(2, 5, 1.0), dense_t=[0.01, 0.05], dense_sample=(np.linspace(0.0, 0.25, 23), 2, 5, 1.0))
E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, tgrid=tg, fixed_diffusion=2)  # fixedMAP
E.emul_solve(vf.rhs_id, 3, 5, True, vf.u0[None, :], vf.p, tgrid=tg, smooth=True)
pl = orc.vector_field("pleiades")
E.emul_solve(pl.rhs_id, 28, 2, True, pl.u0[None, :], pl.p, team="tiles", tgrid=np.arange(4) * 2.0**-10, smooth=True)
fh = orc.vector_field("fhn")
E.emul_solve(fh.rhs_id, 2, 1, False, fh.u0[None, :], fh.p, tgrid=np.arange(9) * 0.07, smooth=True)
print("sanitized emulation run: clean")
PY
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/odef_san_run.py
