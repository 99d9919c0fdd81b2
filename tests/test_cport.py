"""The C restatement used as CPU baseline (oracle/odefilter_cport.c) against the numpy oracle."""
import subprocess
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cport():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import odefilter_cport

    return odefilter_cport


@pytest.mark.parametrize("rhs,q,ek1,dt,t1", [("lorenz63", 3, True, 2.0**-9, 0.5), ("fhn", 1, False, 7e-2, 7.0), ("lotka_volterra", 4, True, 5e-3, 0.5)])
def test_cport_matches_oracle(cport, orc, rhs, q, ek1, dt, t1):
    vf = orc.vector_field(rhs)
    tg = orc.fixed_time_grid(0.0, t1, dt)
    u0s = orc.ensemble_u0(vf.u0, 3, 1e-2)
    mean, cov, _ = cport.filter_fixed(rhs, q, ek1, u0s, vf.p, tg, nthreads=2)
    for i in range(3):
        alg = orc.Alg("EK1" if ek1 else "EK0", q, "dynamic", False)
        ref = orc.solve(vf, alg, u0=u0s[i], tgrid=tg, tspan=(0.0, t1))
        np.testing.assert_allclose(mean[i][: vf.d], ref.x_filt[-1].mu[: vf.d], rtol=1e-11)
        np.testing.assert_allclose(mean[i], ref.x_filt[-1].mu, rtol=1e-5, atol=1e-8)
        c = ref.x_filt[-1].cov()
        assert np.abs(cov[i] - c).max() <= 1e-4 * np.abs(c).max()
