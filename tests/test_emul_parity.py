"""The per-lane device source (odefilters.jl_amd/csrc/ek_lane.h), compiled for the host by
tests/emul, against the oracle.  Catches arithmetic bugs in the kernel source without a GPU.
The GPU tests (test_gpu_parity.py) repeat these through the C ABI on the real kernels."""
import os

import numpy as np
import pytest

import _emul as E
import _parity as P

orc = E.orc

CASES = [
    # rhs, alg, dt, tspan
    ("lorenz63", orc.EK1(order=3), 2.0**-9, (0.0, 0.5)),
    ("fhn", orc.EK0(order=1), 7e-2, (0.0, 20.0)),
    ("fhn", orc.EK1(order=3), 5e-3, (0.0, 1.0)),
    ("lotka_volterra", orc.EK1(order=5), 5e-3, (0.0, 0.5)),
    ("lotka_volterra", orc.EK0(order=2), 5e-3, (0.0, 1.0)),
    ("vanderpol", orc.EK1(order=4), 1e-2, (0.0, 1.0)),
    ("linear", orc.EK0(order=4), 1e-2, (0.0, 1.0)),
]


@pytest.mark.parametrize("rhs,alg,dt,tspan", CASES, ids=[f"{c[0]}-{c[1].kind}{c[1].order}" for c in CASES])
def test_fixed_step_filter_and_smoother(rhs, alg, dt, tspan):
    vf = orc.vector_field(rhs)
    kw = dict(tspan=tspan, dt=dt)
    for smoothed in (False, True):
        base, nm, nc = P.oracle_noise(vf, alg, vf.u0, kw, smoothed)
        r = E.emul_solve(vf.rhs_id, vf.d, alg.order, alg.kind == "EK1", vf.u0[None, :], vf.p, tgrid=np.array(base.t), smooth=True)
        mean, cov = (r["smean"][0], r["scov"][0]) if smoothed else (r["mean"][0], r["cov"][0])
        P.check_against_oracle(mean, cov, base.means(smoothed=smoothed), base.covs(smoothed=smoothed), vf.d, nm, nc,
                               f"{rhs} {alg.kind}({alg.order}) smoothed={smoothed}")
    np.testing.assert_allclose(r["diff"][0][1:], base.diffusions, rtol=max(1e-9, 200 * nc))
    np.testing.assert_allclose(r["loglik"][0], base.log_likelihood, rtol=1e-6)
    assert r["retcode"][0] == 0 and r["naccept"][0] == len(base.t) - 1


@pytest.mark.parametrize("model", ["fixed", "fixedMAP"])
def test_fixed_diffusion(model):
    """FixedDiffusion / MAPFixedDiffusion (src/diffusions.jl:11-36, 46-68) in the lane step."""
    vf = orc.vector_field("lotka_volterra")
    alg = orc.EK1(order=2, diffusionmodel=model, smooth=False)
    sol = orc.solve(vf, alg, dt=5e-3, tspan=(0.0, 0.5))
    r = E.emul_solve(vf.rhs_id, vf.d, 2, True, vf.u0[None, :], vf.p, tgrid=np.array(sol.t), fixed_diffusion={"fixed": 1, "fixedMAP": 2}[model])
    # the lane code returns unscaled covariances + the running-mean diffusion; the postamble rescale
    # (integrator_utils.jl:4-18) is a separate kernel in api.hip, re-done here
    final = r["diff"][0][-1]
    np.testing.assert_allclose(final, sol.diffusions[-1], rtol=1e-9)
    np.testing.assert_allclose(r["mean"][0][:, :2], sol.means()[:, :2], rtol=1e-10)
    assert P.cov_err(r["cov"][0] * final, sol.covs()) < 1e-7


def test_adaptive_matches_oracle_step_sequence():
    """Same PI controller restatement on both sides -> identical accept/reject sequence here."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3, smooth=True)
    sol = orc.solve(vf, alg, adaptive=True, dt=2.0**-9, tspan=(0.0, 0.5))
    r = E.emul_solve(vf.rhs_id, vf.d, 3, True, vf.u0[None, :], vf.p, adaptive=True, t0=0.0, t1=0.5, dt0=2.0**-9,
                     max_save=512, smooth=True)
    n = r["nsaved"][0]
    assert n == len(sol.t) and r["nreject"][0] == sol.nreject and r["retcode"][0] == 0
    np.testing.assert_allclose(r["tsave"][0][:n], sol.t, rtol=1e-9)
    np.testing.assert_allclose(r["mean"][0][:n, :3], sol.means(smoothed=False)[:, :3], rtol=1e-7)
    np.testing.assert_allclose(r["smean"][0][:n, :3], sol.means(smoothed=True)[:, :3], rtol=1e-7)


def test_lagged_record_stores_give_the_same_records():
    """The small-ensemble variant of the every-step filter stores record n while step n + 1 runs (LaggedSink): same bits."""
    vf = orc.vector_field("lorenz63")
    u0s = orc.ensemble_u0(vf.u0, 3, 1e-2)
    tg = np.arange(41) * 2.0**-8
    a = E.emul_solve(vf.rhs_id, 3, 3, True, u0s, vf.p, tgrid=tg, everystep=1)
    b = E.emul_solve(vf.rhs_id, 3, 3, True, u0s, vf.p, tgrid=tg, everystep=2)
    for key in ("mean", "cov", "diff", "loglik"):
        np.testing.assert_array_equal(a[key], b[key])


def test_ensemble_lanes_are_independent():
    """Lane i of a batch equals the single-trajectory run of u0_i (bitwise)."""
    vf = orc.vector_field("lorenz63")
    u0s = orc.ensemble_u0(vf.u0, 5, 1e-2)
    tg = np.arange(33) * 2.0**-9
    rb = E.emul_solve(vf.rhs_id, 3, 3, True, u0s, vf.p, tgrid=tg)
    for i in (0, 4):
        r1 = E.emul_solve(vf.rhs_id, 3, 3, True, u0s[i : i + 1], vf.p, tgrid=tg)
        np.testing.assert_array_equal(rb["mean"][i], r1["mean"][0])
        np.testing.assert_array_equal(rb["cov"][i], r1["cov"][0])


def test_final_only_save_mode():
    vf = orc.vector_field("fhn")
    tg = orc.fixed_time_grid(0.0, 1.0, 7e-2)
    ra = E.emul_solve(vf.rhs_id, 2, 2, True, vf.u0[None, :], vf.p, tgrid=tg, everystep=True)
    rf = E.emul_solve(vf.rhs_id, 2, 2, True, vf.u0[None, :], vf.p, tgrid=tg, everystep=False)
    np.testing.assert_array_equal(ra["mean"][0][-1], rf["mean"][0][0])
    np.testing.assert_array_equal(ra["cov"][0][-1], rf["cov"][0][0])


def test_team_filter_matches_lane_filter_and_oracle():
    """The workgroup-per-trajectory code path (filter_team.h, TEAM = 1 here) on Lorenz-63 against the
    lane-per-trajectory path and the oracle."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3)
    tg = np.arange(129) * 2.0**-9
    base, nm, nc = P.oracle_noise(vf, alg, vf.u0, dict(tspan=(0.0, tg[-1]), dt=2.0**-9), False)
    rt = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, team=True, tgrid=tg)
    rl = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, tgrid=tg)
    P.check_against_oracle(rt["mean"][0], rt["cov"][0], base.means(smoothed=False), base.covs(smoothed=False), 3, nm, nc, "team")
    np.testing.assert_allclose(rt["mean"][0][:, :3], rl["mean"][0][:, :3], rtol=1e-12)
    np.testing.assert_allclose(rt["loglik"], rl["loglik"], rtol=1e-9)


@pytest.mark.parametrize("path", [True, "tiles"])
@pytest.mark.parametrize("q,ek1", [(2, True), (3, False), (5, True)])
def test_pleiades_team_filter(q, ek1, path):
    """BASELINE config 4 problem (d = 28) through the team path: Taylor-mode init with r^-3 jets, filter, smoother."""
    vf = orc.vector_field("pleiades")
    alg = orc.Alg("EK1" if ek1 else "EK0", q, "dynamic", True)
    ns = 10
    sol = orc.solve(vf, alg, dt=2.0**-10, tspan=(0.0, ns * 2.0**-10))
    r = E.emul_solve(vf.rhs_id, 28, q, ek1, vf.u0[None, :], vf.p, team=path, tgrid=np.array(sol.t), smooth=True)
    M = sol.means(smoothed=False)
    np.testing.assert_allclose(r["mean"][0][0], M[0], rtol=1e-13, atol=1e-13)  # Taylor-mode initial state
    np.testing.assert_allclose(r["mean"][0][:, :28], M[:, :28], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(r["smean"][0][:, :28], sol.means(smoothed=True)[:, :28], rtol=1e-12, atol=1e-13)
    assert r["retcode"][0] == 0
    if q < 5:  # at order 5 the residual of the first steps is pure rounding noise in BOTH implementations
        assert P.cov_err(r["cov"][0], sol.covs(smoothed=False)) < 1e-6


@pytest.mark.parametrize("ek1", [True, False])
def test_pleiades_fixed_diffusion_tiles(ek1):
    """FixedDiffusion (src/diffusions.jl:11-36, static order of src/perform_step.jl:56-63) on the tiled path: the
    running mean of z'S^-1 z / d is kept by the helper wavefront."""
    vf = orc.vector_field("pleiades")
    alg = orc.Alg("EK1" if ek1 else "EK0", 2, "fixed", False)
    ns = 8
    sol = orc.solve(vf, alg, dt=2.0**-10, tspan=(0.0, ns * 2.0**-10))
    r = E.emul_solve(vf.rhs_id, 28, 2, ek1, vf.u0[None, :], vf.p, team="tiles", tgrid=np.array(sol.t), fixed_diffusion=True)
    np.testing.assert_allclose(r["mean"][0][:, :28], sol.means(smoothed=False)[:, :28], rtol=1e-12, atol=1e-13)
    # the kernel records the running mean per step; the reference overwrites all entries with the final value in its
    # postamble (src/integrator_utils.jl:4-18), which the library does in a separate rescale pass
    np.testing.assert_allclose(r["diff"][0][-1], sol.diffusions[-1], rtol=1e-9)


@pytest.mark.parametrize("q,ek1", [(2, True), (3, False)])
def test_pleiades_adaptive_tiles(q, ek1):
    """Adaptive stepping on the tiled workgroup path (TilesFilter::run_adaptive): same accept/reject sequence and
    posterior as the oracle's OrdinaryDiffEq loop, then the team smoother over the per-attempt records."""
    vf = orc.vector_field("pleiades")
    alg = orc.Alg("EK1" if ek1 else "EK0", q, "dynamic", True)
    kw = dict(abstol=1e-8, reltol=1e-6)
    dt0 = 0.02  # far too large a first step: 4-5 rejected attempts before the controller settles
    sol = orc.solve(vf, alg, adaptive=True, dt=dt0, tspan=(0.0, 0.05), **kw)
    assert sol.nreject >= 3
    r = E.emul_solve(vf.rhs_id, 28, q, ek1, vf.u0[None, :], vf.p, team="tiles", adaptive=True, t0=0.0, t1=0.05, dt0=dt0,
                     max_save=256, smooth=True, **kw)
    n = r["nsaved"][0]
    assert r["retcode"][0] == 0 and n == len(sol.t) and r["nreject"][0] == sol.nreject, (n, len(sol.t), r["nreject"], sol.nreject)
    assert r["nsaved_raw"][0] == n + sol.nreject  # one device record per attempted step
    # step sizes come out of a (q+1)-th root of an error estimate that sits near rounding level relative to the state
    np.testing.assert_allclose(r["tsave"][0][:n], sol.t, rtol=1e-6)
    np.testing.assert_allclose(r["mean"][0][:n, :28], sol.means(smoothed=False)[:, :28], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(r["smean"][0][:n, :28], sol.means(smoothed=True)[:, :28], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("q", [4, 5])
def test_rows_smoother_larger_state(q):
    """D = 15 / 18 (Lorenz, order 4 / 5): the row-per-lane team smoother (smooth_rows.h, 16- and 32-lane teams)."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=q)
    kw = dict(tspan=(0.0, 0.125), dt=2.0**-8)
    base, nm, nc = P.oracle_noise(vf, alg, vf.u0, kw, True)
    r = E.emul_solve(vf.rhs_id, 3, q, True, vf.u0[None, :], vf.p, tgrid=np.array(base.t), smooth=True)
    P.check_against_oracle(r["smean"][0], r["scov"][0], base.means(smoothed=True), base.covs(smoothed=True), 3, nm, nc, f"rows q={q}")


@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("smoothed", [False, True])
def test_dense_output(adaptive, smoothed):
    """sol(t) (src/solution.jl:165-210): interior points, exact grid points, and beyond the last time."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3, smooth=smoothed)
    if adaptive:
        sol = orc.solve(vf, alg, adaptive=True, dt=2.0**-9, tspan=(0.0, 0.5))
        kw = dict(adaptive=True, t0=0.0, t1=0.5, dt0=2.0**-9, max_save=256)
    else:
        sol = orc.solve(vf, alg, dt=2.0**-6, tspan=(0.0, 0.5))
        kw = dict(tgrid=np.array(sol.t))
    # exact stored times are only bit-equal between the two sides on the fixed grid (adaptive times agree to 1e-16)
    exact = [sol.t[3], sol.t[-2]] if not adaptive else []
    tq = np.concatenate([np.linspace(0.0, 0.5, 23), exact, [0.5, 0.51, 0.6]])
    r = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, smooth=smoothed, dense_t=tq, **kw)
    consts = orc.make_consts(3, 3)
    for j, t in enumerate(tq):
        ref = orc.dense_output(sol, consts, float(t), smoothed=smoothed)
        np.testing.assert_allclose(r["qmean"][0][j][:3], ref.mu[:3], rtol=1e-7, atol=1e-10, err_msg=f"t={t}")
        np.testing.assert_allclose(r["qmean"][0][j], ref.mu, rtol=1e-5, atol=1e-6, err_msg=f"t={t}")
        c = ref.cov()
        assert np.abs(r["qcov"][0][j] - c).max() <= 1e-6 * np.abs(c).max() + 1e-300, f"t={t}"


@pytest.mark.parametrize("adaptive", [False, True])
def test_posterior_sampling(adaptive):
    """sample_states (src/solution_sampling.jl:24-62) of the device source against the oracle with the same N(0,1)
    stream and the same (lower-triangular) square root; the zero-noise chain reproduces the smoothed means."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3)
    if adaptive:
        sol = orc.solve(vf, alg, adaptive=True, dt=2.0**-9, tspan=(0.0, 0.5))
        kw = dict(adaptive=True, t0=0.0, t1=0.5, dt0=2.0**-9, max_save=256)
    else:
        sol = orc.solve(vf, alg, dt=2.0**-6, tspan=(0.0, 0.5))
        kw = dict(tgrid=np.array(sol.t))
    consts = orc.make_consts(3, 3)
    seed, n = 1234, 3
    r = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, smooth=True, sample=(n, seed, 1.0), **kw)
    ns, cap = len(sol.t), r["samples"].shape[1]
    S = r["samples"][0][:ns]
    raw = r["raw_index"][0] if adaptive else np.arange(ns)  # device slot of the k-th accepted record
    # a rejected attempt repeats the record; the draw of a repeated state is made at its LAST repeat
    raw = np.append(raw[1:] - 1, raw[-1])
    ref = orc.sample_states(sol, consts, n, sqrt="cholesky",
                            normal=lambda j, slot, k: orc.sample_normal(seed, 0, j, int(raw[slot]), k, n, cap, 12))
    scale = np.abs(ref).max(axis=(0, 2))[None, :, None]
    err = (np.abs(S - ref) / scale).max(axis=(0, 2))
    assert err[:3].max() < 1e-8 and err.max() < 1e-4, err  # u block / ill-conditioned derivative blocks
    r0 = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, smooth=True, sample=(1, seed, 0.0), **kw)
    np.testing.assert_allclose(r0["samples"][0][1:ns, :3, 0], sol.means(smoothed=True)[1:, :3], rtol=1e-7)


@pytest.mark.parametrize("adaptive", [False, True])
def test_dense_posterior_sampling(adaptive):
    """dense_sample_states (src/solution_sampling.jl:63-69): filter posterior interpolated at a dense grid, then the
    same backward sampler with the diffusion of an interval looked up by time (:41); device source against the oracle
    with the same N(0,1) stream and square root."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3)
    if adaptive:
        sol = orc.solve(vf, alg, adaptive=True, dt=2.0**-9, tspan=(0.0, 0.5))
        kw = dict(adaptive=True, t0=0.0, t1=0.5, dt0=2.0**-9, max_save=256)
    else:
        sol = orc.solve(vf, alg, dt=2.0**-6, tspan=(0.0, 0.5))
        kw = dict(tgrid=np.array(sol.t))
    consts = orc.make_consts(3, 3)
    seed, n = 99, 2
    times = np.linspace(sol.t[0], sol.t[-1], 57)  # off-grid and (first, last) on-grid times
    r = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, smooth=True, dense_sample=(times, n, seed, 1.0), **kw)
    S = r["dense_samples"][0]
    assert S.shape == (len(times), 12, n)
    ref, tt = orc.dense_sample_states(sol, consts, n, times=times, sqrt="cholesky", seed=seed)
    np.testing.assert_array_equal(tt, times)
    scale = np.abs(ref).max(axis=(0, 2))[None, :, None]
    err = (np.abs(S - ref) / scale).max(axis=(0, 2))
    assert err[:3].max() < 1e-8 and err.max() < 1e-4, err


def test_dense_sampling_edge_grids():
    """Degenerate dense grids: a single time (only the draw from the filter marginal), repeated times (h = 0: the
    state is the later one, as for duplicated save times, src/smoothing.jl:13-16) and a grid equal to the solver's own
    (then dense_sample_states is sample_states, both with the reference's square root replaced by the lower factor)."""
    vf = orc.vector_field("lorenz63")
    sol = orc.solve(vf, orc.EK1(order=3), dt=2.0**-6, tspan=(0.0, 0.25))
    consts = orc.make_consts(3, 3)
    kw = dict(tgrid=np.array(sol.t), smooth=True)
    one = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, dense_sample=([0.1], 2, 3, 1.0), **kw)["dense_samples"][0]
    ref, _ = orc.dense_sample_states(sol, consts, 2, times=[0.1], sqrt="cholesky", seed=3)
    np.testing.assert_allclose(one[:, :3], ref[:, :3], rtol=1e-9)
    tq = [0.05, 0.1, 0.1, 0.2]
    dup = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, dense_sample=(tq, 2, 3, 1.0), **kw)["dense_samples"][0]
    np.testing.assert_array_equal(dup[1], dup[2])
    assert np.isfinite(dup).all()
    grid = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, dense_sample=(sol.t, 2, 3, 1.0), sample=(2, 3, 1.0), **kw)
    scale = np.abs(grid["samples"][0]).max(axis=(0, 2))[None, :, None]
    assert (np.abs(grid["dense_samples"][0][1:] - grid["samples"][0][1:]) / scale).max() < 1e-9


ROWS_CASES = [c for c in CASES if orc.vector_field(c[0]).d * (c[1].order + 1) <= 16] + [
    ("lorenz63", orc.EK1(order=4), 2.0**-8, (0.0, 0.25)),  # D = 15: three idle lanes per team... one, of 16
    ("lorenz63", orc.EK0(order=1), 2.0**-8, (0.0, 0.25)),
    # stiff, order 5: without the symmetrisation exchange at the end of the step (P7) the rounding-level antisymmetric part
    # of the lanes' rows grew exponentially here (covariance off by 7e-5 after 50 steps, 2e-3 after smoothing)
    ("vanderpol", orc.EK1(order=5), 2e-2, (0.0, 1.0)),
]


@pytest.mark.parametrize("rhs,alg,dt,tspan", ROWS_CASES, ids=[f"{c[0]}-{c[1].kind}{c[1].order}" for c in ROWS_CASES])
def test_row_team_filter(rhs, alg, dt, tspan):
    """filter_rows.h (16 lanes per trajectory, the small-ensemble filter): same checks as the lane filter, every-step
    and final-only records, an ensemble of three so that teams and trajectories are told apart."""
    vf = orc.vector_field(rhs)
    kw = dict(tspan=tspan, dt=dt)
    base, nm, nc = P.oracle_noise(vf, alg, vf.u0, kw, False)
    u0s = np.stack([vf.u0, vf.u0 * (1 + 1e-3), vf.u0])
    r = E.emul_solve(vf.rhs_id, vf.d, alg.order, alg.kind == "EK1", u0s, vf.p, tgrid=np.array(base.t), everystep=3, smooth=True)
    sbase, snm, snc = P.oracle_noise(vf, alg, vf.u0, kw, True)
    for i in (0, 2):
        P.check_against_oracle(r["mean"][i], r["cov"][i], base.means(smoothed=False), base.covs(smoothed=False), vf.d, nm, nc,
                               f"rows {rhs} {alg.kind}({alg.order}) traj {i}")
        P.check_against_oracle(r["smean"][i], r["scov"][i], sbase.means(smoothed=True), sbase.covs(smoothed=True), vf.d, snm, snc,
                               f"rows + smoother {rhs} {alg.kind}({alg.order}) traj {i}")
        np.testing.assert_allclose(r["diff"][i][1:], base.diffusions, rtol=max(1e-9, 200 * nc))
        np.testing.assert_allclose(r["loglik"][i], base.log_likelihood, rtol=1e-6)
    assert np.abs(r["mean"][1] - r["mean"][0]).max() > 0  # the perturbed trajectory is a different one
    assert (r["retcode"] == 0).all() and (r["naccept"] == len(base.t) - 1).all() and (r["nsaved"] == len(base.t)).all()
    f = E.emul_solve(vf.rhs_id, vf.d, alg.order, alg.kind == "EK1", u0s, vf.p, tgrid=np.array(base.t), everystep=-1)
    np.testing.assert_array_equal(f["mean"][:, 0], r["mean"][:, -1])
    np.testing.assert_array_equal(f["cov"][:, 0], r["cov"][:, -1])
    np.testing.assert_array_equal(f["diff"][:, 0], r["diff"][:, -1])
    assert (f["nsaved"] == 1).all()


@pytest.mark.parametrize("model", ["fixed", "fixedMAP"])
def test_row_team_filter_static_diffusion(model):
    vf = orc.vector_field("lotka_volterra")
    sol = orc.solve(vf, orc.EK1(order=2, diffusionmodel=model, smooth=False), dt=5e-3, tspan=(0.0, 0.5))
    r = E.emul_solve(vf.rhs_id, vf.d, 2, True, vf.u0[None, :], vf.p, tgrid=np.array(sol.t), everystep=3,
                     fixed_diffusion={"fixed": 1, "fixedMAP": 2}[model])
    final = r["diff"][0][-1]
    np.testing.assert_allclose(final, sol.diffusions[-1], rtol=1e-9)
    np.testing.assert_allclose(r["mean"][0][:, :2], sol.means()[:, :2], rtol=1e-10)
    assert P.cov_err(r["cov"][0] * final, sol.covs()) < 1e-7


@pytest.mark.parametrize("rhs,order,t1", [("lorenz63", 3, 0.5), ("fhn", 2, 2.0), ("vanderpol", 4, 0.3)])
def test_row_team_adaptive_filter_and_smoother(rhs, order, t1):
    """rows_filter.h / rows_smooth.h, adaptive: the PI-controller loop of a 16-lane team (one record per ATTEMPTED step,
    rejected attempts repeat the old state) and the backward pass over those records, against the oracle's
    OrdinaryDiffEq restatement -- same accepted / rejected step sequence, posterior at the oracle's noise level."""
    vf = orc.vector_field(rhs)
    alg = orc.EK1(order=order, smooth=True)
    sol = orc.solve(vf, alg, adaptive=True, dt=2.0**-9, tspan=(0.0, t1))
    u0s = np.stack([vf.u0, vf.u0 * (1 + 1e-3)])
    r = E.emul_solve(vf.rhs_id, vf.d, order, True, u0s, vf.p, adaptive=True, t0=0.0, t1=t1, dt0=2.0**-9, max_save=1024,
                     smooth=True, everystep=3)
    n = r["nsaved"][0]
    assert n == len(sol.t) and r["nreject"][0] == sol.nreject and r["retcode"][0] == 0
    assert r["nsaved_raw"][0] == sol.naccept + sol.nreject + 1  # one record per attempt
    # (stiff van der Pol at order 4: the error estimate itself carries the 1e-8 noise of the higher derivatives, _parity.py)
    rt = 1e-6 if rhs == "vanderpol" else 1e-9
    np.testing.assert_allclose(r["tsave"][0][:n], sol.t, rtol=rt)
    np.testing.assert_allclose(r["mean"][0][:n, :vf.d], sol.means(smoothed=False)[:, :vf.d], rtol=100 * rt)
    np.testing.assert_allclose(r["smean"][0][:n, :vf.d], sol.means(smoothed=True)[:, :vf.d], rtol=100 * rt)
    np.testing.assert_allclose(r["diff"][0][1:n], sol.diffusions, rtol=1e-4)
    np.testing.assert_allclose(r["loglik"][0], sol.log_likelihood, rtol=1e-6)
    assert P.cov_err(r["scov"][0][:n], sol.covs(smoothed=True)) < 1e-5
    # the lane kernels on the same problem: same step sequence, same posterior to rounding
    l = E.emul_solve(vf.rhs_id, vf.d, order, True, u0s, vf.p, adaptive=True, t0=0.0, t1=t1, dt0=2.0**-9, max_save=1024, smooth=True)
    assert (l["nsaved"] == r["nsaved"]).all() and (l["nreject"] == r["nreject"]).all()
    np.testing.assert_allclose(r["smean"][1][: r["nsaved"][1], :vf.d], l["smean"][1][: l["nsaved"][1], :vf.d], rtol=100 * rt)


def test_row_team_adaptive_rejections_and_limits():
    """A first step that is far too long is rejected (the record repeats the initial state at t0); a record budget that is
    too small ends in MaxIters with what was saved; static diffusion rides along."""
    vf = orc.vector_field("lorenz63")
    alg = orc.EK1(order=3, smooth=True)
    sol = orc.solve(vf, alg, adaptive=True, dt=0.25, tspan=(0.0, 0.5))
    assert sol.nreject >= 1
    r = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, adaptive=True, t0=0.0, t1=0.5, dt0=0.25, max_save=512,
                     smooth=True, everystep=3)
    n = r["nsaved"][0]
    assert n == len(sol.t) and r["nreject"][0] == sol.nreject and r["naccept"][0] == sol.naccept
    np.testing.assert_allclose(r["smean"][0][:n, :3], sol.means(smoothed=True)[:, :3], rtol=1e-7)
    short = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, adaptive=True, t0=0.0, t1=0.5, dt0=2.0**-9, max_save=8,
                         everystep=3)
    assert short["retcode"][0] == 1 and short["nsaved_raw"][0] == 8
    fx = orc.solve(vf, orc.EK1(order=3, diffusionmodel="fixed", smooth=False), adaptive=True, dt=2.0**-9, tspan=(0.0, 0.25))
    rf = E.emul_solve(vf.rhs_id, 3, 3, True, vf.u0[None, :], vf.p, adaptive=True, t0=0.0, t1=0.25, dt0=2.0**-9, max_save=512,
                      everystep=3, fixed_diffusion=1)
    nf = rf["nsaved"][0]
    assert nf == len(fx.t)
    np.testing.assert_allclose(rf["mean"][0][:nf, :3], fx.means(smoothed=False)[:, :3], rtol=1e-7)


@pytest.mark.parametrize("kernel", ["lane", "lagged", "rows"])
def test_zero_pivots_and_zero_reflector_norms(kernel):
    """The Cholesky-failure branch of the reference (src/filtering.jl:38-47) at its extreme: u' = 0 (linear field with
    p = 0) has z = 0 exactly, hence sigma^2 = 0 and a ZERO predicted covariance -- every Cholesky pivot and every
    reflector norm of the step is 0.  The reference's QR fallback yields a zero factor and then stops in inv(S) of a
    singular S; the kernels' zero-pivot rule (column zeroed, reciprocal taken as 0) must return the exact constant
    solution with zero covariance and no NaN (round 1 left rsq(0) * 0 = NaN in the reflector norm)."""
    vf = orc.vector_field("linear")
    u0 = np.array([[0.75, -1.25], [2.0, 3.0]])
    tg = np.arange(17) * 2.0**-6
    ev = {"lane": True, "lagged": 2, "rows": 3}[kernel]
    for q in (1, 3):
        r = E.emul_solve(vf.rhs_id, 2, q, True, u0, np.zeros(2), tgrid=tg, everystep=ev, smooth=True)
        assert (r["retcode"] == 0).all()
        np.testing.assert_array_equal(r["mean"][:, :, :2], np.broadcast_to(u0[:, None, :], (2, 17, 2)))
        assert np.all(r["mean"][:, :, 2:] == 0.0) and np.all(r["cov"] == 0.0) and np.all(r["diff"] == 0.0)
        np.testing.assert_array_equal(r["smean"][:, :, :2], r["mean"][:, :, :2])
        assert np.all(r["scov"] == 0.0)


# ---- against EXACT evaluations of the reference algorithm (tests/golden/make_exact.py) --------------------------------

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("kernel", ["lane", "rows"])
def test_lorenz_1024_steps_against_50_digit_evaluation(kernel):
    """BASELINE configs 2/3 (trajectory 0, all 1 024 steps), filter and smoother: the kernels' arithmetic (run on the host)
    is as close to the 50-digit mpmath evaluation of the reference algorithm as the float64 oracle is, block by block."""
    fx = np.load(os.path.join(GOLD, "exact_lorenz_mp.npz"))
    vf = orc.vector_field("lorenz63")
    tg = np.arange(int(fx["nsteps"]) + 1) * float(fx["dt"])
    r = E.emul_solve(vf.rhs_id, 3, 3, True, fx["u0"][None, :], vf.p, tgrid=tg, everystep={"lane": True, "rows": 3}[kernel], smooth=True)
    P.check_against_exact(r["mean"][0], r["cov"][0], fx["mean_filt"], fx["cov_filt"], fx["oracle_block_err_filt"],
                          fx["oracle_cov_err_filt"], 3, f"{kernel} filter")
    P.check_against_exact(r["smean"][0], r["scov"][0], fx["mean_smooth"], fx["cov_smooth"], fx["oracle_block_err_smooth"],
                          fx["oracle_cov_err_smooth"], 3, f"{kernel} smoother")
    np.testing.assert_allclose(r["diff"][0][1:], fx["diffusions"], rtol=2e-3)  # a squared residual: noise of u''' twice over


def test_pleiades_24_steps_against_extended_precision():
    """BASELINE config 4 at test size: the tiled D = 168 filter against an x87-extended evaluation."""
    fx = np.load(os.path.join(GOLD, "exact_pleiades_ld.npz"))
    vf = orc.vector_field("pleiades")
    tg = np.arange(int(fx["nsteps"]) + 1) * float(fx["dt"])
    r = E.emul_solve(vf.rhs_id, 28, 5, True, fx["u0"][None, :], vf.p, team="tiles", tgrid=tg)
    be = P.block_err(r["mean"][0], fx["mean_filt"], 28)
    assert be[0] <= P.U_RTOL
    assert np.all(be <= np.maximum(P.EXACT_FACTOR * fx["oracle_block_err_filt"], 1e-15)), (be, fx["oracle_block_err_filt"])
    assert P.cov_err(r["cov"][0][-1:], fx["cov_final"][None]) <= P.EXACT_FACTOR * float(fx["oracle_cov_err_final"])
