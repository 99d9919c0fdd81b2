"""Parity of the HIP kernels (through the C ABI, libodefilter_hip.so) with the oracle and the
committed golden fixtures.  Needs a real MI355X: run with `-m gpu`."""
import os
import sys

import numpy as np
import pytest

import _parity as P
import odefilter_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _alg(pkg, kind, order, diffusion="dynamic", smooth=True):
    return (pkg.EK1 if kind == "EK1" else pkg.EK0)(order=order, diffusionmodel=diffusion, smooth=smooth)


# ---- golden fixtures ------------------------------------------------------------------------

GOLDEN_FIXED = ["fhn_ek0_q1_cfg1", "lorenz_ek1_q3", "lv_ek1_q2_fixeddiff", "lv_ek0_q4", "vdp_ek1_q5"]


# The fixed-step filter has a second, experimental mapping for D <= 16 (csrc/filter_rows.h, 16 lanes per trajectory; off by
# default: it measured no faster than the lane filter); ODEF_FILTER_ROWS_MAX_N is read at every launch and
# selects it for ensembles below that size.
FILTER_KERNELS = {"lane": "0", "rows": "1000000000"}


@pytest.mark.parametrize("kernel", ["lane", "rows"])
@pytest.mark.parametrize("name", GOLDEN_FIXED)
def test_golden_fixed_step(pkg, name, kernel, monkeypatch):
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", FILTER_KERNELS[kernel])
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rhs, kind, order = str(g["rhs"]), str(g["kind"]), int(g["order"])
    diffusion = str(g["diffusionmodel"])
    vf = orc.vector_field(rhs)
    u0s = g["u0s"]
    prob = pkg.EnsembleProblem(pkg.ODEProblem(rhs, u0s[0], tuple(g["tspan"]), g["p"]), u0s=u0s)
    sol = pkg.solve(prob, _alg(pkg, kind, order, diffusion), pkg.EnsembleHIP(), dt=float(g["dt"]), adaptive=False)
    assert sol.retcode == ["Success"] * len(u0s)
    np.testing.assert_array_equal(sol.t, g["t"][0])
    alg_o = orc.Alg(kind, order, diffusion, True)
    kw = dict(tspan=tuple(g["tspan"]), dt=float(g["dt"]))
    mf, cf, ms, cs = sol.x_filt_mean(), sol.x_filt_cov(), sol.x_smooth_mean(), sol.x_smooth_cov()
    for i in range(len(u0s)):
        for smoothed, (m, c, gm, gc) in ((False, (mf, cf, g["mean_filt"], g["cov_filt"])),
                                         (True, (ms, cs, g["mean_smooth"], g["cov_smooth"]))):
            _, nm, nc = P.oracle_noise(vf, alg_o, u0s[i], kw, smoothed)
            P.check_against_oracle(m[i], c[i], gm[i], gc[i], vf.d, nm, nc, f"{name}[{i}] smoothed={smoothed}")
    np.testing.assert_allclose(sol.diffusions, g["diffusions"], rtol=1e-6)
    if diffusion == "dynamic":
        np.testing.assert_allclose(sol.log_likelihood, g["loglik"], rtol=1e-6)
    else:
        assert np.all(np.isnan(sol.log_likelihood))  # src/integrator_utils.jl:6
    # sol.pu[1]: mean u0, zero covariance (test/solution.jl:38-41)
    np.testing.assert_array_equal(sol.x_filt_mean()[:, 0, : vf.d], u0s)
    assert np.all(sol.x_filt_cov()[:, 0] == 0.0)


def test_golden_adaptive(pkg):
    """Config-5 shape: adaptive PI + RTS.  Step sequences coincide because both sides restate the
    same controller; parity is asserted at the solver tolerance on common times as well."""
    g = np.load(os.path.join(GOLD, "lorenz_ek1_q3_adaptive.npz"))
    u0s = g["u0s"]
    prob = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", u0s[0], tuple(g["tspan"]), g["p"]), u0s=u0s)
    sol = pkg.solve(prob, pkg.EK1(order=3), pkg.EnsembleHIP(), dt=float(g["dt"]), adaptive=True,
                    abstol=float(g["abstol"]), reltol=float(g["reltol"]), max_steps=256)
    assert sol.retcode == ["Success"] * 3
    np.testing.assert_array_equal(sol.destats.naccept, g["naccept"])
    np.testing.assert_array_equal(sol.destats.nreject, g["nreject"])
    for i in range(3):
        n = int(sol.nsaved[i])
        assert n == g["naccept"][i] + 1
        np.testing.assert_allclose(sol.t[i, :n], g["t"][i, :n], rtol=1e-9)
        np.testing.assert_allclose(sol.x_filt_mean()[i, :n, :3], g["mean_filt"][i, :n, :3], rtol=1e-7)
        np.testing.assert_allclose(sol.u[i, :n], g["mean_smooth"][i, :n, :3], rtol=1e-7)
        assert sol.t[i, n - 1] == g["tspan"][1]


# ---- seeded parity against the oracle at sizes it finishes in seconds ------------------------


@pytest.mark.parametrize("rhs,kind,q,dt,t1", [
    ("lorenz63", "EK1", 3, 2.0**-9, 2.0),      # BASELINE config 2/3 problem, full time span
    ("lorenz63", "EK0", 2, 2.0**-9, 0.25),
    ("lorenz63", "EK1", 5, 2.0**-8, 0.25),
    ("fhn", "EK1", 4, 5e-3, 1.0),
    ("lotka_volterra", "EK0", 1, 5e-3, 1.0),
    ("vanderpol", "EK0", 3, 1e-2, 1.0),
    ("linear", "EK1", 2, 1e-2, 1.0),
])
@pytest.mark.parametrize("kernel", ["lane", "rows"])
def test_ensemble_parity_with_oracle(pkg, rhs, kind, q, dt, t1, kernel, monkeypatch):
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", FILTER_KERNELS[kernel])
    vf = orc.vector_field(rhs)
    N = 130  # ragged: not a multiple of the 64-lane wavefront
    ens = pkg.EnsembleProblem(pkg.ODEProblem(rhs, vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, _alg(pkg, kind, q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    np.testing.assert_array_equal(sol.ctx.get(13).T, u0s)  # device splitmix64 ensemble == oracle's, bit for bit
    assert sol.retcode == ["Success"] * N
    alg_o = orc.Alg(kind, q, "dynamic", True)
    mf, cf, ms, cs = sol.x_filt_mean(), sol.x_filt_cov(), sol.x_smooth_mean(), sol.x_smooth_cov()
    for i in (0, 63, 64, 129):
        for smoothed, (m, c) in ((False, (mf, cf)), (True, (ms, cs))):
            base, nm, nc = P.oracle_noise(vf, alg_o, u0s[i], dict(tspan=(0.0, t1), dt=dt), smoothed)
            P.check_against_oracle(m[i], c[i], base.means(smoothed=smoothed), base.covs(smoothed=smoothed), vf.d, nm, nc,
                                   f"{rhs} {kind}({q}) traj {i} smoothed={smoothed}")


def test_row_team_filter_equals_lane_filter(pkg, monkeypatch):
    """The two fixed-step filter kernels on the same ragged ensemble (every-step and final-only records, per-trajectory
    parameters): results agree to rounding, counters and return codes are identical."""
    vf = orc.vector_field("lorenz63")
    N, t1, dt = 203, 0.5, 2.0**-8
    rng = np.random.default_rng(5)
    ps = np.asarray(vf.p)[None, :] * (1 + 1e-3 * rng.standard_normal((N, 3)))
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), u0s=u0s, ps=ps)
    out = {}
    for kernel, v in FILTER_KERNELS.items():
        monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", v)
        for every in (True, False):
            sol = pkg.solve(ens, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False, save_everystep=every)
            out[kernel, every] = (sol.x_filt_mean().copy(), sol.x_filt_cov().copy(), sol.log_likelihood.copy(),
                                  sol.destats.nf.copy(), sol.destats.njacs.copy(), list(sol.retcode), np.asarray(sol.nsaved).copy())
    for every in (True, False):
        a, b = out["lane", every], out["rows", every]
        scale = np.abs(a[0]).max(axis=(0, 1))
        assert (np.abs(a[0] - b[0]).max(axis=(0, 1)) / scale)[:3].max() < 1e-11
        assert (np.abs(a[0] - b[0]).max(axis=(0, 1)) / scale).max() < 1e-7  # derivative blocks (ill-conditioned)
        assert P.cov_err(b[1], a[1]) < 1e-5
        np.testing.assert_allclose(b[2], a[2], rtol=1e-8)
        for k in (3, 4, 6):
            np.testing.assert_array_equal(a[k], b[k])
        assert a[5] == b[5] == ["Success"] * N
    np.testing.assert_array_equal(out["rows", False][0][:, 0], out["rows", True][0][:, -1])  # final-only = last record


@pytest.mark.parametrize("adaptive", [False, True])
def test_both_small_state_smoothers_against_oracle(pkg, adaptive, monkeypatch):
    """D <= 12 has three smoother kernels, chosen by ensemble size (csrc/ek_kernels.h LaunchSmooth): DPP row teams for small
    ensembles, LDS row teams, one lane per trajectory for large ones.  The environment switches the launcher reads at every
    launch force each in turn."""
    vf = orc.vector_field("lorenz63")
    N, t1 = 130, 0.5
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    kw = dict(dt=2.0**-9, adaptive=True, max_steps=256) if adaptive else dict(dt=2.0**-7, adaptive=False)
    sols = {}
    kernels = {"lane": ("0", "1"), "rows": ("0", "1000000000"), "bcast": ("1000000000", "1")}
    for name, (rows_max, lane_min) in kernels.items():
        monkeypatch.setenv("ODEF_SMOOTH_ROWS_MAX_N", rows_max)
        monkeypatch.setenv("ODEF_SMOOTH_LANE_MIN_N", lane_min)
        sols[name] = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, **kw)
        assert sols[name].retcode == ["Success"] * N
        _ = sols[name].x_smooth_mean()  # fetch while the override is in place (lazy accessors)
        _ = sols[name].x_smooth_cov()
    for i in (0, 64, 129):
        ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), **({"dt": 2.0**-9, "adaptive": True} if adaptive else {"dt": 2.0**-7}))
        n = len(ref.t)
        for name in kernels:
            m, c = sols[name].x_smooth_mean()[i][:n], sols[name].x_smooth_cov()[i][:n]
            np.testing.assert_allclose(m[:, :3], ref.means(smoothed=True)[:, :3], rtol=1e-7 if adaptive else 1e-10, err_msg=name)
            assert P.cov_err(c, ref.covs(smoothed=True)) < 1e-5, name
    for name in ("rows", "bcast"):
        np.testing.assert_allclose(sols[name].x_smooth_mean()[..., :3], sols["lane"].x_smooth_mean()[..., :3], rtol=1e-9, atol=1e-12, err_msg=name)


@pytest.mark.parametrize("rhs,order", [("fhn", 1), ("fhn", 3), ("lotka_volterra", 2), ("vanderpol", 4), ("lotka_volterra", 5), ("lorenz63", 1), ("lorenz63", 2)])
def test_lane_smoother_other_state_dimensions(pkg, rhs, order, monkeypatch):
    """The one-lane-per-trajectory smoother (smooth_lane.h: hand-managed AGPR file, two result rows per pass) at D = 4, 6, 8,
    9, 10, 12 (d = 2 and 3; odd D has an unpaired last row), an ensemble that does not fill its last wavefront: against
    the oracle and against the LDS row-team smoother on the same filter records."""
    vf = orc.vector_field(rhs)
    N, dt, t1 = 77, 2.0**-7, 0.5
    monkeypatch.setenv("ODEF_SMOOTH_ROWS_MAX_N", "0")
    ens = pkg.EnsembleProblem(pkg.ODEProblem(rhs, vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    out = {}
    for name, lane_min in (("lane", "1"), ("rows", "1000000000")):
        monkeypatch.setenv("ODEF_SMOOTH_LANE_MIN_N", lane_min)
        sol = pkg.solve(ens, pkg.EK1(order=order), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
        assert sol.retcode == ["Success"] * N
        out[name] = (sol.x_smooth_mean(), sol.x_smooth_cov(), sol.x_filt_mean(), sol.x_filt_cov())
    pm, pc, fm, fc = out["lane"]
    lm = out["rows"][0]
    np.testing.assert_array_equal(pm[:, [0, -1]], fm[:, [0, -1]])  # first and last record are copied (src/smoothing.jl:11)
    np.testing.assert_array_equal(pc[:, [0, -1]], fc[:, [0, -1]])
    for i in (0, 31, 32, 76):
        alg = orc.EK1(order=order)
        base, nm, nc = P.oracle_noise(vf, alg, u0s[i], dict(tspan=(0.0, t1), dt=dt), True)
        P.check_against_oracle(pm[i], pc[i], base.means(smoothed=True), base.covs(smoothed=True), vf.d, nm, nc, f"lane smoother {rhs}({order}) traj {i}")
        np.testing.assert_allclose(pm[i][:, : vf.d], lm[i][:, : vf.d], rtol=1e-10)


def test_per_trajectory_parameters(pkg):
    vf = orc.vector_field("lotka_volterra")
    N = 70
    rng = np.random.default_rng(3)
    ps = vf.p[None, :] * (1.0 + 0.05 * rng.standard_normal((N, 4)))
    u0s = np.tile(vf.u0, (N, 1))
    prob = pkg.EnsembleProblem(pkg.ODEProblem("lotka_volterra", vf.u0, (0.0, 0.5), vf.p), u0s=u0s, ps=ps)
    sol = pkg.solve(prob, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=5e-3, adaptive=False)
    for i in (0, 69):
        ref = orc.solve(vf, orc.EK1(order=3, smooth=False), p=ps[i], tspan=(0.0, 0.5), dt=5e-3)
        np.testing.assert_allclose(sol.u[i], ref.u, rtol=1e-10)


# ---- step-level entry points: test/filtering.jl identities through the C ABI -----------------


def _rand_lower(rng, n, d):
    return np.tril(rng.random((n, d, d)))


def test_predict_update_smooth_step_identities(pkg):
    """test/filtering.jl:28-46, 79-89, 118-122 on a seeded batch (d = 5, o = 3)."""
    rng = np.random.default_rng(11)
    n, d, o = 67, 5, 3
    m, L = rng.random((n, d)), _rand_lower(rng, n, d)
    A, LQ = rng.random((d, d)), np.tril(rng.random((d, d)))
    Pm = L @ L.transpose(0, 2, 1)
    mo, co = pkg.predict(m, L, A, LQ)
    np.testing.assert_allclose(mo, m @ A.T, rtol=1e-14)
    np.testing.assert_allclose(co, A @ Pm @ A.T + LQ @ LQ.T, rtol=1e-8)
    H = rng.random((n, o, d))
    z = np.einsum("nod,nd->no", H, m)
    S = H @ Pm @ H.transpose(0, 2, 1)
    K = Pm @ H.transpose(0, 2, 1) @ np.linalg.inv(S)
    mu, cu = pkg.update(m, L, H, z)
    np.testing.assert_allclose(mu, m + np.einsum("ndo,no->nd", K, 0 - z), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cu, Pm - K @ S @ K.transpose(0, 2, 1), rtol=1e-6, atol=1e-9)
    ms_, Ls = rng.random((n, d)), _rand_lower(rng, n, d)
    Ps = Ls @ Ls.transpose(0, 2, 1)
    mp_, Pp = m @ A.T, A @ Pm @ A.T + LQ @ LQ.T
    G = Pm @ A.T @ np.linalg.inv(Pp)
    msm, csm = pkg.smooth_step(m, L, ms_, Ls, A, LQ)
    np.testing.assert_allclose(msm, m + np.einsum("nij,nj->ni", G, ms_ - mp_), rtol=1e-7)
    np.testing.assert_allclose(csm, Pm + G @ (Ps - Pp) @ G.transpose(0, 2, 1), rtol=1e-6, atol=1e-8)
    # against the oracle functions themselves
    for i in (0, 66):
        xo = orc.predict(orc.SRGaussian(m[i], L[i]), A, LQ)
        np.testing.assert_allclose(co[i], xo.cov(), rtol=1e-12)
        xs, _ = orc.smooth(orc.SRGaussian(m[i], L[i]), orc.SRGaussian(ms_[i], Ls[i]), A, LQ)
        np.testing.assert_allclose(csm[i], xs.cov(), rtol=1e-9, atol=1e-12)


def test_empty_batch_and_bad_dims(pkg):
    A = np.eye(3)
    mo, co = pkg.predict(np.zeros((0, 3)), np.zeros((0, 3, 3)), A, A)
    assert mo.shape == (0, 3) and co.shape == (0, 3, 3)
    with pytest.raises(pkg.OdefError):
        pkg.predict(np.zeros((1, 40)), np.zeros((1, 40, 40)), np.eye(40), np.eye(40))  # > ODEF_MAX_STEP_DIM


# ---- BASELINE-size runs: size-independent properties ------------------------------------------


def test_full_size_properties(pkg, monkeypatch):
    """65 536 trajectories (BASELINE config 3 ensemble), shortened time span so the output fits a
    test: (a) a sample of trajectories matches the oracle, (b) duplicated inputs give bitwise equal
    outputs wherever they sit in the batch (same kernel; the row-team kernel a small ensemble gets by default agrees to
    rounding), (c) covariances are PSD with exact zero initial block, (d) final-only save mode equals the last
    every-step record."""
    vf = orc.vector_field("lorenz63")
    N, nsteps, dt = 65536, 32, 2.0**-9
    tspan = (0.0, nsteps * dt)
    u0s = orc.ensemble_u0(vf.u0, 8, 1e-2)
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, tspan, vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    assert np.all(sol.retcode_raw == 0)
    mean = sol.ctx.get(0)  # [n_save, D, N]
    cov = sol.ctx.get(1)
    for i in range(8):
        ref = orc.solve(vf, orc.EK1(order=3, smooth=False), u0=u0s[i], tspan=tspan, dt=dt)
        np.testing.assert_allclose(mean[:, :3, i], ref.u, rtol=1e-11)
    Cl = pkg.unpack_tril(cov[-1].T, 12)
    w = np.linalg.eigvalsh(Cl[:4096])
    assert w.min() > -1e-9 * np.abs(w).max()
    assert np.all(cov[0] == 0.0)
    # (b) permutation / position independence
    pick = np.array([5, 70, 4097, 65535])
    u0_dev = sol.ctx.get(13).T
    prob2 = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, tspan, vf.p), u0s=u0_dev[pick][::-1].copy())
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", "0")  # the lane kernel, as for the 65 536
    sol2 = pkg.solve(prob2, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False)
    np.testing.assert_array_equal(sol2.ctx.get(0)[:, :, ::-1], mean[:, :, pick])
    np.testing.assert_array_equal(sol2.ctx.get(1)[:, :, ::-1], cov[:, :, pick])
    monkeypatch.delenv("ODEF_FILTER_ROWS_MAX_N")  # default: 4 trajectories go to the row-team kernel
    sol2r = pkg.solve(prob2, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False)
    np.testing.assert_allclose(sol2r.ctx.get(0)[:, :3, ::-1], mean[:, :3, pick], rtol=1e-11)
    # (d) final-only
    sol3 = pkg.solve(ens, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False,
                     save_everystep=False)
    np.testing.assert_array_equal(sol3.ctx.get(0)[0], mean[-1])
    np.testing.assert_array_equal(sol3.ctx.get(1)[0], cov[-1])


def test_repeated_solves_reuse_and_replace_the_grid_tables(pkg):
    """odef_solve_fixed keeps the preconditioner tables of the last grid on the device and skips the uploads when the
    same grid comes again; a different grid (same length, other steps), an adaptive solve in between, and the first grid
    again must each give exactly what a fresh context gives."""
    vf = orc.vector_field("lorenz63")
    N = 70
    grid_a = np.arange(33) * 2.0**-8
    grid_b = np.cumsum(np.concatenate([[0.0], np.tile([2.0**-9, 2.0**-8], 16)]))  # same length, alternating steps

    def fresh(grid):
        c = pkg.Context("lorenz63", 3, 1, N, smooth=False)
        c.set_problem_perturbed(list(vf.u0), list(vf.p), 0.0, 1e-2)
        c.solve_fixed(grid)
        out = (c.get(0).copy(), c.get(1).copy(), c.get(2).copy())
        c.close()
        return out

    want_a, want_b = fresh(grid_a), fresh(grid_b)
    c = pkg.Context("lorenz63", 3, 1, N, smooth=False)
    c.set_problem_perturbed(list(vf.u0), list(vf.p), 0.0, 1e-2)
    for grid, want in ((grid_a, want_a), (grid_a, want_a), (grid_b, want_b), (grid_a, want_a)):
        c.solve_fixed(grid)
        for f in range(3):
            np.testing.assert_array_equal(c.get(f), want[f])
    c.solve_adaptive(float(grid_a[-1]), 1e-6, 1e-3, 2.0**-9, None, 64)
    c.solve_fixed(grid_a)
    for f in range(3):
        np.testing.assert_array_equal(c.get(f), want_a[f])
    c.close()


def test_caller_owned_output_buffers_and_stream(pkg):
    """odef_bind_device / odef_set_stream with torch-owned memory and torch's stream."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("torch's bundled HIP runtime does not see the GPU in this process (the library itself is exercised by every other test)")
    vf = orc.vector_field("lorenz63")
    N, nsteps, dt = 256, 16, 2.0**-9
    ctx = pkg.Context("lorenz63", 3, 1, N, save_everystep=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    out = torch.zeros(12, N, dtype=torch.float64, device="cuda")
    ctx.bind_device(0, out.data_ptr(), out.numel() * 8)
    ctx.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
    ctx.solve_fixed(np.arange(nsteps + 1) * dt)
    torch.cuda.synchronize()
    ref = orc.solve(vf, orc.EK1(order=3, smooth=False), u0=orc.ensemble_u0(vf.u0, 1, 1e-2)[0], tspan=(0.0, nsteps * dt), dt=dt)
    np.testing.assert_allclose(out[:3, 0].cpu().numpy(), ref.u[-1], rtol=1e-12)
    ptr, nbytes = ctx.device_ptr(0)
    assert ptr == out.data_ptr() and nbytes == 12 * N * 8
    ms, nl = ctx.kernel_time_ms(0)
    assert ms > 0 and nl == 1
    ctx.close()


# ---- BASELINE config 4: Pleiades, d = 28, EK1(order=5), D = 168 (workgroup-per-trajectory kernels) ----


def test_pleiades_config4_golden(pkg):
    g = np.load(os.path.join(GOLD, "pleiades_ek1_q5_cfg4.npz"))
    vf = orc.vector_field("pleiades")
    u0s, ns, dt = g["u0s"], int(g["nsteps"]), float(g["dt"])
    prob = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", u0s[0], (0.0, ns * dt), ()), u0s=u0s)
    sol = pkg.solve(prob, pkg.EK1(order=5, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] * len(u0s)
    np.testing.assert_allclose(sol.u, g["u"], rtol=1e-11, atol=1e-13)
    m = sol.x_filt_mean()
    be = P.block_err(m[:, -1], g["mean_final"], 28)
    assert be[0] < 1e-11 and be[1] < 1e-9, be
    # final-only save mode gives the same last record
    sol2 = pkg.solve(prob, pkg.EK1(order=5, smooth=False), pkg.EnsembleHIP(), dt=dt, adaptive=False, save_everystep=False)
    np.testing.assert_array_equal(sol2.x_filt_mean()[:, 0], m[:, -1])
    np.testing.assert_array_equal(sol2.x_filt_cov()[:, 0], sol.x_filt_cov()[:, -1])


PLEIADES_EXACT = [(2, "EK1", ""), (3, "EK0", ""), (5, "EK1", ""), (5, "EK1", "_dt6")]


@pytest.mark.parametrize("kernels", ["mfma+split", "mfma+persistent", "tiles+split"])
@pytest.mark.parametrize("q,kind,tag", PLEIADES_EXACT, ids=[f"{k}{q}{t}" for q, k, t in PLEIADES_EXACT])
def test_pleiades_ensemble_parity(pkg, q, kind, tag, kernels, monkeypatch):
    """The workgroup-per-trajectory kernels (D = 84, 112, 168) against EXTENDED-PRECISION evaluations of the reference
    algorithm (tests/golden/exact_pleiades_*_smooth_ld.npz, generator make_exact.py): perturbed positions (1e-3, first 14
    components) as in SURVEY 8(d) config 4, 12 steps, filter + smoother, trajectories 0 and 4 of the ensemble.  Every
    derivative block of the filtered and smoothed means, and the covariance of the last filter record and of smoothed
    record 1, must be as close to the exact result as the float64 oracle is (factor 16) -- for the matrix-core filter
    (Joseph form) and the register-tile filter (square-root form), the split smoother (a kernel per phase, sweeps on chip)
    and the persistent one.  BASELINE order 5 twice: at config 4's dt = 2^-10, where the first residuals are h^5-small and the
    reference arithmetic itself is rounding noise in the top blocks (the yardstick says so), and at dt = 2^-6, where the
    same kernels are pinned to 1e-6 in the covariance.  No tolerance here is calibrated on a float64 run."""
    fx = np.load(os.path.join(GOLD, f"exact_pleiades_{kind.lower()}q{q}{tag}_smooth_ld.npz"))
    filt, smoother = kernels.split("+")
    monkeypatch.setenv("ODEF_PLEIADES_FILTER", "tiles" if filt == "tiles" else "")
    monkeypatch.setenv("ODEF_SMOOTH_SPLIT", "1" if smoother == "split" else "0")
    vf = orc.vector_field("pleiades")
    N, ns, dt = 5, int(fx["nsteps"]), float(fx["dt"])
    ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, ns * dt), ()), perturb_scale=1e-3, n_perturbed=14)
    sol = pkg.solve(ens, _alg(pkg, kind, q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-3, n_perturbed=14)
    np.testing.assert_array_equal(sol.ctx.get(13).T, u0s)
    assert sol.retcode == ["Success"] * N
    mf, ms, cf, cs = sol.x_filt_mean(), sol.x_smooth_mean(), sol.x_filt_cov(), sol.x_smooth_cov()
    for k, i in enumerate(fx["trajs"]):
        np.testing.assert_array_equal(u0s[i], fx["u0s"][k])
        np.testing.assert_allclose(mf[i][:, :28], fx["mean_filt"][k][:, :28], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(ms[i][:, :28], fx["mean_smooth"][k][:, :28], rtol=1e-11, atol=1e-13)
        P.check_against_exact_fixture(fx, k, mf[i], cf[i], ms[i], cs[i], 28, f"pleiades {kind}({q}){tag} {kernels} traj {i}")
        # positive semi-definite, on the scale of the largest eigenvalue
        for c in (cf[i][-1], cs[i][1]):
            w = np.linalg.eigvalsh(c)
            assert w.min() >= -1e-9 * np.abs(w).max()


# ---- the workgroup-per-trajectory kernels on a second shape: Lorenz-96, d = 16 (csrc/inst_lorenz96.hip) ----


@pytest.mark.parametrize("smoother", ["split", "persistent"])
@pytest.mark.parametrize("q,kind", [(2, "EK1"), (3, "EK1"), (3, "EK0"), (5, "EK1")])
def test_lorenz96_on_the_matrix_core_kernels(pkg, q, kind, smoother, monkeypatch):
    """filter_mfma.h / smooth_mfma.h are templates over the vector field; until round 3 only Pleiades (d = 28) instantiated
    them.  Lorenz-96 with 16 variables runs the same kernels with derivative blocks of two 8-row tiles (state dimension 48,
    64, 96) and WITHOUT the team-evaluation hooks Pleiades provides (f and the Jacobian come from the generic one-lane path):
    filter and smoother against the oracle, means of the solution at 1e-10, the higher blocks and the covariances at the
    oracle's own rounding-noise level, positive semi-definite covariances."""
    monkeypatch.setenv("ODEF_SMOOTH_SPLIT", "1" if smoother == "split" else "0")
    vf = orc.vector_field("lorenz96")
    N, ns, dt = 6, 12, 2.0**-7
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz96", vf.u0, (0.0, ns * dt), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, _alg(pkg, kind, q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] * N
    assert "ek_filter_mfma_kernel<odef::RhsLorenz96" in sol.ctx.kernel_name(0)
    assert ("rts_smooth_sweeps_kernel<16" if smoother == "split" else "rts_smooth_mfma_kernel<16") in sol.ctx.kernel_name(1)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    np.testing.assert_array_equal(sol.ctx.get(13).T, u0s)
    alg_o = orc.Alg(kind, q, "dynamic", True)
    mf, ms, cf, cs = sol.x_filt_mean(), sol.x_smooth_mean(), sol.x_filt_cov(), sol.x_smooth_cov()
    for i in (0, 5):
        for smoothed, m, c in ((False, mf, cf), (True, ms, cs)):
            base, nm, nc = P.oracle_noise(vf, alg_o, u0s[i], dict(tspan=(0.0, ns * dt), dt=dt), smoothed)
            P.check_against_oracle(m[i], c[i], base.means(smoothed=smoothed), base.covs(smoothed=smoothed), 16, nm, nc,
                                   f"lorenz96 {kind}({q}) {smoother} traj {i} smoothed={smoothed}")
        for c in (cf[i][-1], cs[i][1]):
            w = np.linalg.eigvalsh(c)
            assert w.min() >= -1e-9 * np.abs(w).max()
    np.testing.assert_allclose(sol.log_likelihood[0], orc.solve(vf, alg_o, u0=u0s[0], tspan=(0.0, ns * dt), dt=dt).log_likelihood, rtol=1e-6)


def test_lorenz96_adaptive_dense_output_and_sampling(pkg):
    """... and the rest of the path on that shape: the adaptive matrix-core filter (PI controller, one record per attempt)
    against the oracle's OrdinaryDiffEq loop, the smoother over its records, sol(t) (csrc/dense_mfma.h) and posterior
    sampling (csrc/sample_mfma.h) at D = 64."""
    vf = orc.vector_field("lorenz96")
    N, t1 = 3, 0.1
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz96", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-8, adaptive=True, abstol=1e-8, reltol=1e-6, max_steps=256)
    assert sol.retcode == ["Success"] * N
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(16, 3)
    for i in (0, 2):
        ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-8, adaptive=True, abstol=1e-8, reltol=1e-6)
        n = len(ref.t)
        assert int(sol.nsaved[i]) == n and int(sol.destats.nreject[i]) == ref.nreject
        np.testing.assert_allclose(sol.t[i][:n], ref.t, rtol=1e-6)
        np.testing.assert_allclose(sol.u[i][:n], ref.u, rtol=1e-6, atol=1e-9)
        tq = np.array([0.011, 0.05, t1])
        qm, qc = sol(tq)
        want = np.array([orc.dense_output(ref, consts, float(t)).mu[:16] for t in tq])
        np.testing.assert_allclose(qm[i][:, :16], want, rtol=1e-6, atol=1e-9)
    smp = sol.sample_states(2, 11, noise_scale=0.0)  # the chain of conditional means = the smoothed means
    n0 = int(sol.nsaved[0])
    np.testing.assert_allclose(smp[0][:n0, :16, 0], sol.x_smooth_mean()[0][:n0, :16], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("q", [1, 3, 5])
def test_pleiades_mfma_kernel_against_tiles_kernel_nonuniform_grid(pkg, q, monkeypatch):
    """The two D = 28 (q+1) fixed-step filters -- matrix cores / Joseph form (csrc/filter_mfma.h, default) and register tiles /
    square-root form (csrc/filter_tiles.h) -- on a grid whose step size changes (the MFMA kernel's helper rebuilds its
    coefficient tables and LDS preconditioner table only then), every step saved: solution block to 1e-11, all records finite,
    same diffusions to 1e-6, same log-likelihood to 1e-6 relative."""
    vf = orc.vector_field("pleiades")
    N = 3
    grid = np.concatenate([np.arange(6) * 2.0**-10, 5 * 2.0**-10 + np.arange(1, 5) * 2.0**-11, [5 * 2.0**-10 + 4 * 2.0**-11 + 2.0**-9]])
    out = {}
    for name, env in (("mfma", ""), ("tiles", "tiles")):
        monkeypatch.setenv("ODEF_PLEIADES_FILTER", env)
        ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
        ctx.set_problem_perturbed(vf.u0, [], 0.0, 1e-3, n_perturbed=14)
        ctx.solve_fixed(grid)
        assert (ctx.get(10) == 0).all()
        out[name] = (ctx.get(0).copy(), ctx.get(1).copy(), ctx.get(2).copy(), ctx.get(4).copy())
        ctx.close()
    (m1, c1, d1, l1), (m0, c0, d0, l0) = out["mfma"], out["tiles"]
    assert np.isfinite(m1).all() and np.isfinite(c1).all()
    np.testing.assert_allclose(m1[:, :28], m0[:, :28], rtol=1e-11, atol=1e-13)
    # Order 5 at these step sizes: the residuals of the first steps are h^5-small, i.e. rounding noise in ANY float64 arithmetic
    # (tests/golden/exact_pleiades_ek1q5_smooth_ld.npz: the oracle's own diffusions are 40 % - 300x off the extended-precision
    # ones), so two correct kernels need not agree in them; what order 5 guarantees is asserted against the extended-precision
    # fixtures in test_pleiades_ensemble_parity (both filters, dt = 2^-10 and 2^-6).
    np.testing.assert_allclose(d1[2:], d0[2:], rtol=1e-6 if q < 5 else 0.2)
    if q < 5:
        np.testing.assert_allclose(l1, l0, rtol=1e-6)


@pytest.mark.parametrize("q", [2, 5])
def test_pleiades_smoother_record_stage(pkg, q, monkeypatch):
    """The D = 28 (q+1) smoother on a fixed grid reads and writes its covariance records through a trajectory-major stage
    (csrc/record_stage.h), in as many chunks as the stage budget needs.  Only addresses change: the whole stage, a stage of
    a few records (N = 70 is not a multiple of the 64-wide transposition tiles; several launches with the carried state in the
    workspace) and the records in place (budget 0) must agree bit for bit (a grid with two step sizes)."""
    vf = orc.vector_field("pleiades")
    N, dt = 70, 2.0**-10
    grid = np.concatenate([np.arange(8) * dt, 7 * dt + np.arange(1, 7) * dt / 2])  # 14 records
    D = 28 * (q + 1)
    per_rec_mb = N * ((D * (D + 1) // 2 + 15) // 16 * 16) * 8 / 2**20
    out = {}
    monkeypatch.setenv("ODEF_SMOOTH_SPLIT", "0")  # one persistent launch per block, the same arithmetic as the in-place pass
    for name, mb in (("whole", None), ("chunks", str(int(np.ceil(3.2 * per_rec_mb)))), ("in place", "0"), ("split", None),
                     ("split chunks", str(int(np.ceil(3.2 * per_rec_mb))))):
        if mb is None:
            monkeypatch.delenv("ODEF_SMOOTH_STAGE_MB", raising=False)
        else:
            monkeypatch.setenv("ODEF_SMOOTH_STAGE_MB", mb)
        monkeypatch.setenv("ODEF_SMOOTH_SPLIT", "1" if name.startswith("split") else "0")
        ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
        ctx.set_problem_perturbed(vf.u0, [], 0.0, 1e-3, n_perturbed=14)
        ctx.solve_fixed(grid)
        ctx.smooth()
        assert (ctx.get(10) == 0).all()
        out[name] = (ctx.get(11).copy(), ctx.get(12).copy(), ctx.get(0).copy(), ctx.get(1).copy())
        ctx.close()
    assert np.isfinite(out["whole"][0]).all() and np.isfinite(out["whole"][1]).all()
    for name in ("chunks", "in place"):
        np.testing.assert_array_equal(out[name][0], out["whole"][0], err_msg=name)
        np.testing.assert_array_equal(out[name][1], out["whole"][1], err_msg=name)
        # the filter in front of it writes its records through the same stage when all of them fit ("whole" only)
        np.testing.assert_array_equal(out[name][2], out["whole"][2], err_msg=name)
        np.testing.assert_array_equal(out[name][3], out["whole"][3], err_msg=name)
    # the split pass (default: a kernel per phase and record, the sweeps with the factor in LDS) orders the sweeps' sums
    # differently: same results to rounding, whole stage and blocks bit-identical to each other
    np.testing.assert_array_equal(out["split chunks"][0], out["split"][0])
    np.testing.assert_array_equal(out["split chunks"][1], out["split"][1])
    D_ = 28 * (q + 1)
    sd = np.sqrt(np.maximum(out["whole"][1][:, [k * (k + 1) // 2 + k for k in range(D_)]], 0.0))
    assert (np.abs(out["split"][0] - out["whole"][0]) <= (1e-12 if q < 5 else 1e-4) * (np.abs(out["whole"][0]) + sd) + 1e-300).all()
    scale = np.abs(out["whole"][1]).max(axis=1, keepdims=True)
    assert (np.abs(out["split"][1] - out["whole"][1]) <= (1e-12 if q < 5 else 1e-8) * scale + 1e-300).all()


@pytest.mark.parametrize("kind,q", [("EK1", 2), ("EK0", 3), ("EK1", 5)])
def test_pleiades_adaptive(pkg, kind, q):
    """The reference's default solve is adaptive: PI-controlled steps on the workgroup-per-trajectory path
    (TilesFilter::run_adaptive), one device record per attempted step, team smoother over them.  The first step is far too
    large, so the run starts with rejected attempts."""
    vf = orc.vector_field("pleiades")
    N, t1, dt0 = 5, 0.05, 0.02
    tol = dict(abstol=1e-8, reltol=1e-6)
    ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, t1), ()), perturb_scale=1e-3, n_perturbed=14)
    sol = pkg.solve(ens, _alg(pkg, kind, q), pkg.EnsembleHIP(), trajectories=N, adaptive=True, dt=dt0, max_steps=256, **tol)
    assert sol.retcode == ["Success"] * N
    u0s = orc.ensemble_u0(vf.u0, N, 1e-3, n_perturbed=14)
    for i in (0, 4):
        ref = orc.solve(vf, orc.Alg(kind, q, "dynamic", True), u0=u0s[i], adaptive=True, dt=dt0, tspan=(0.0, t1), **tol)
        n = int(sol.nsaved[i])
        assert (q == 5 or ref.nreject >= 1) and int(sol.destats.nreject[i]) == ref.nreject and n == len(ref.t)
        np.testing.assert_allclose(sol.t[i, :n], ref.t, rtol=1e-6)
        np.testing.assert_allclose(sol.x_filt_mean()[i, :n, :28], ref.means(smoothed=False)[:, :28], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(sol.u[i, :n], ref.u, rtol=1e-7, atol=1e-12)
        assert sol.t[i, n - 1] == t1


@pytest.mark.parametrize("q", [1, 3, 4])
def test_pleiades_smoother_split_pass_orders(pkg, q, monkeypatch):
    """The split pass of the D = 28 (q+1) smoother (a kernel per phase and record; factorisation and sweeps on chip in
    rts_smooth_sweeps_kernel, one wavefront per tile column: 4, 7 and 9 wavefronts here) against the persistent kernel:
    the same algebra in a different summation order.  Orders 2 and 5 are covered by test_pleiades_smoother_record_stage."""
    vf = orc.vector_field("pleiades")
    N = 20
    grid = np.arange(10) * 2.0**-10
    out = {}
    for name, env in (("persistent", "0"), ("split", "1"), ("split, Y' through the workspace", "1")):
        monkeypatch.setenv("ODEF_SMOOTH_SPLIT", env)
        # (default for d = 28: the on-chip kernel forms Y' = A X from the record in registers; =0: the predict kernel hands it over)
        monkeypatch.setenv("ODEF_SMOOTH_YFROMX", "0" if "workspace" in name else "1")
        ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
        ctx.set_problem_perturbed(vf.u0, [], 0.0, 1e-3, n_perturbed=14)
        ctx.solve_fixed(grid)
        ctx.smooth()
        assert (ctx.get(10) == 0).all()
        out[name] = (ctx.get(11).copy(), ctx.get(12).copy())
        ctx.close()
    (m0, c0), (m1, c1) = out["persistent"], out["split"]
    assert np.isfinite(m1).all() and np.isfinite(c1).all()
    # the two ways Y' reaches the on-chip kernel sum the same terms in the same order
    np.testing.assert_array_equal(out["split, Y' through the workspace"][0], m1)
    np.testing.assert_array_equal(out["split, Y' through the workspace"][1], c1)
    D = 28 * (q + 1)
    sd = np.sqrt(np.maximum(c0[:, [k * (k + 1) // 2 + k for k in range(D)]], 0.0))
    tol = {1: 1e-13, 3: 1e-9, 4: 1e-7}[q]  # the higher orders' last derivative blocks are ill-conditioned
    assert (np.abs(m1 - m0) <= tol * (np.abs(m0) + sd) + 1e-300).all()
    assert (np.abs(c1 - c0) <= tol * np.abs(c0).max(axis=1, keepdims=True) + 1e-300).all()
    np.testing.assert_allclose(m1[:, :28], m0[:, :28], rtol=1e-12, atol=1e-15)  # the solution block itself


def test_pleiades_adaptive_smoother_record_stage(pkg, monkeypatch):
    """Adaptive solves give every trajectory its own number of records: the staged smoother pass (csrc/record_stage.h) lets
    each trajectory join in at the block of the stage that holds its last record.  Whole stage, blocks of a few records and
    the in-place pass agree bit for bit, unused save slots stay zero; the tolerances are set so that record counts differ."""
    vf = orc.vector_field("pleiades")
    N, q, t1 = 70, 2, 0.06
    D = 28 * (q + 1)
    per_rec_mb = N * ((D * (D + 1) // 2 + 15) // 16 * 16) * 8 / 2**20
    out = {}
    for name, mb in (("whole", None), ("blocks", str(int(np.ceil(3.2 * per_rec_mb)))), ("in place", "0"), ("split", None),
                     ("split blocks", str(int(np.ceil(3.2 * per_rec_mb))))):
        if mb is None:
            monkeypatch.delenv("ODEF_SMOOTH_STAGE_MB", raising=False)
        else:
            monkeypatch.setenv("ODEF_SMOOTH_STAGE_MB", mb)
        monkeypatch.setenv("ODEF_SMOOTH_SPLIT", "1" if name.startswith("split") else "0")
        ctx = pkg.Context("pleiades", q, 1, N, save_everystep=True)
        ctx.set_problem_perturbed(vf.u0, [], 0.0, 3e-2, n_perturbed=14)
        ctx.solve_adaptive(t1, 1e-7, 1e-5, 0.02, None, 63)
        ctx.smooth()
        assert (ctx.get(10) == 0).all()
        out[name] = (ctx.get(11).copy(), ctx.get(12).copy(), ctx.get(9).copy())
        ctx.close()
    ns = out["whole"][2]
    assert ns.min() >= 3 and ns.max() > ns.min() and ns.max() < out["whole"][1].shape[0]
    sm, sc = out["whole"][0], out["whole"][1]
    assert np.isfinite(sm).all() and np.isfinite(sc).all()
    for i in (0, int(np.argmin(ns)), int(np.argmax(ns))):
        assert (sc[ns[i]:, :, i] == 0).all() and np.abs(sc[ns[i] - 1, :, i]).max() > 0
    for name in ("blocks", "in place"):
        np.testing.assert_array_equal(out[name][2], ns)
        np.testing.assert_array_equal(out[name][0], sm, err_msg=name)
        np.testing.assert_array_equal(out[name][1], sc, err_msg=name)
    # the split pass (a kernel per phase and record; trajectories without the record skip inside the kernels)
    np.testing.assert_array_equal(out["split"][2], ns)
    np.testing.assert_array_equal(out["split blocks"][0], out["split"][0])
    np.testing.assert_array_equal(out["split blocks"][1], out["split"][1])
    for i in (0, int(np.argmin(ns)), int(np.argmax(ns))):
        assert (out["split"][1][ns[i]:, :, i] == 0).all()
    sd = np.sqrt(np.maximum(sc[:, [k * (k + 1) // 2 + k for k in range(D)]], 0.0))
    assert (np.abs(out["split"][0] - sm) <= 1e-12 * (np.abs(sm) + sd) + 1e-300).all()
    assert (np.abs(out["split"][1] - sc) <= 1e-12 * np.abs(sc).max(axis=1, keepdims=True) + 1e-300).all()


@pytest.mark.parametrize("kind,q", [("EK1", 3), ("EK0", 5)])
def test_pleiades_adaptive_mfma_kernel_against_tiles_kernel(pkg, kind, q, monkeypatch):
    """The two adaptive D = 28 (q+1) filters -- matrix cores / Joseph form (MfmaFilter::run_adaptive, default) and register
    tiles / square-root form (TilesFilter::run_adaptive) -- under the same controller: same accepted and rejected attempts,
    same save times to 1e-6 and solution block to 1e-7 (the tolerances of the oracle test above); a run that starts with rejections (records re-read from the previous attempt)."""
    vf = orc.vector_field("pleiades")
    N, t1, dt0 = 6, 0.05, 0.02
    out = {}
    for name, env in (("mfma", ""), ("tiles", "tiles")):
        monkeypatch.setenv("ODEF_PLEIADES_FILTER", env)
        ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, t1), ()), perturb_scale=1e-3, n_perturbed=14)
        sol = pkg.solve(ens, _alg(pkg, kind, q), pkg.EnsembleHIP(), trajectories=N, adaptive=True, dt=dt0, max_steps=256,
                        abstol=1e-8, reltol=1e-6)
        assert sol.retcode == ["Success"] * N
        out[name] = (np.array(sol.nsaved), np.array(sol.destats.nreject), np.array(sol.destats.naccept), sol.t.copy(),
                     sol.x_filt_mean().copy(), sol.x_filt_cov().copy())
    a, b = out["mfma"], out["tiles"]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    assert (q == 5 or (a[1] >= 1).all()) and np.isfinite(a[5]).all()
    for i in range(N):
        n = int(a[0][i])
        # the step sizes come out of pow(error estimate): the two covariance forms differ in the last digits of it
        np.testing.assert_allclose(a[3][i, :n], b[3][i, :n], rtol=1e-6)
        np.testing.assert_allclose(a[4][i, :n, :28], b[4][i, :n, :28], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("model", ["fixed", "fixedMAP"])
@pytest.mark.parametrize("adaptive", [False, True])
def test_static_diffusion_models(pkg, model, adaptive):
    """FixedDiffusion / MAPFixedDiffusion (src/diffusions.jl:11-36, 46-68; static order of src/perform_step.jl:56-63)
    with the post-hoc calibration of postamble! (src/integrator_utils.jl:4-18), fixed grid and adaptive stepping (the
    running estimate then only advances on accepted steps, and the rescale ends at each trajectory's own last record)."""
    vf = orc.vector_field("lotka_volterra")
    N, t1 = 70, 0.5
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lotka_volterra", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    kw = dict(adaptive=True, dt=2.0**-8) if adaptive else dict(adaptive=False, dt=2.0**-6)
    sol = pkg.solve(ens, pkg.EK1(order=3, diffusionmodel=model), pkg.EnsembleHIP(), trajectories=N, **kw)
    assert sol.retcode == ["Success"] * N
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    for i in (0, 69):
        ref = orc.solve(vf, orc.Alg("EK1", 3, model, True), u0=u0s[i], tspan=(0.0, t1), **kw)
        n = len(ref.t)
        assert int(sol.nsaved[i]) == n
        np.testing.assert_allclose(sol.u[i, :n], ref.u, rtol=1e-6 if adaptive else 1e-10, atol=1e-12)
        np.testing.assert_allclose(sol.diffusions[i, : n - 1], ref.diffusions, rtol=1e-5 if adaptive else 1e-8)
        assert P.cov_err(sol.x_filt_cov()[i, :n], ref.covs(smoothed=False)) < (1e-4 if adaptive else 1e-6)
        assert P.cov_err(sol.x_smooth_cov()[i, :n], ref.covs(smoothed=True)) < (1e-4 if adaptive else 1e-6)
    assert np.isnan(sol.log_likelihood).all()  # src/integrator_utils.jl:7


@pytest.mark.parametrize("model", ["fixed", "fixedMAP"])
def test_pleiades_fixed_diffusion(pkg, model):
    """FixedDiffusion / MAPFixedDiffusion on the workgroup-per-trajectory path (tiled filter + helper wavefront, post-hoc
    covariance rescale, team smoother): src/diffusions.jl:11-36, 46-68, src/integrator_utils.jl:4-18."""
    vf = orc.vector_field("pleiades")
    ns, dt = 8, 2.0**-10
    prob = pkg.ODEProblem("pleiades", vf.u0, (0.0, ns * dt), ())
    sol = pkg.solve(prob, pkg.EK1(order=2, diffusionmodel=model), dt=dt, adaptive=False)
    ref = orc.solve(vf, orc.Alg("EK1", 2, model, True), dt=dt, tspan=(0.0, ns * dt))
    np.testing.assert_allclose(sol.u[0], ref.u, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(sol.diffusions[0], ref.diffusions, rtol=1e-8)
    assert P.cov_err(sol.x_filt_cov()[0], ref.covs(smoothed=False)) < 1e-6
    assert P.cov_err(sol.x_smooth_cov()[0], ref.covs(smoothed=True)) < 1e-6


@pytest.mark.parametrize("model", ["fixed", "fixedMAP"])
def test_pleiades_fixed_diffusion_adaptive(pkg, model):
    """The static diffusion models under the adaptive controller on the matrix-core kernel (MfmaFilter::run_adaptive): the
    running estimate advances on accepted attempts only and is put back on a rejected one, the error estimate uses the local
    diffusion of the attempt; against the oracle's loop, with rejected attempts in the run."""
    vf = orc.vector_field("pleiades")
    t1, dt0 = 0.04, 0.02
    tol = dict(abstol=1e-8, reltol=1e-6)
    prob = pkg.ODEProblem("pleiades", vf.u0, (0.0, t1), ())
    sol = pkg.solve(prob, pkg.EK1(order=2, diffusionmodel=model), adaptive=True, dt=dt0, max_steps=256, **tol)
    assert sol.retcode == ["Success"]
    ref = orc.solve(vf, orc.Alg("EK1", 2, model, True), adaptive=True, dt=dt0, tspan=(0.0, t1), **tol)
    n = int(sol.nsaved[0])
    assert ref.nreject >= 1 and int(sol.destats.nreject[0]) == ref.nreject and n == len(ref.t)
    np.testing.assert_allclose(sol.t[0, :n], ref.t, rtol=1e-6)
    np.testing.assert_allclose(sol.u[0, :n], ref.u, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(sol.diffusions[0, : n - 1], ref.diffusions, rtol=1e-5)


# ---- dense output / saveat (src/solution.jl:165-210) -----------------------------------------------


@pytest.mark.parametrize("smooth", [False, True])
def test_dense_output_fixed_grid(pkg, smooth):
    vf = orc.vector_field("lorenz63")
    N, dt, t1 = 70, 2.0**-6, 0.5
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=3, smooth=smooth), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    grid = sol.t
    tq = np.concatenate([np.linspace(0.0, t1, 17), [grid[3], grid[-2], t1, t1 + 0.01]])
    qm, qc = sol(tq)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, 3)
    for i in (0, 69):
        ref = orc.solve(vf, orc.EK1(order=3, smooth=smooth), u0=u0s[i], tspan=(0.0, t1), dt=dt)
        for j, t in enumerate(tq):
            r = orc.dense_output(ref, consts, float(t), smoothed=smooth)
            np.testing.assert_allclose(qm[i, j, :3], r.mu[:3], rtol=1e-9, atol=1e-12, err_msg=f"traj {i} t={t}")
            c = r.cov()
            assert np.abs(qc[i, j] - c).max() <= 1e-6 * np.abs(c).max() + 1e-300
    # at stored times the dense output is the stored record itself
    k = 3
    np.testing.assert_array_equal(qm[:, 17], (sol.x_smooth_mean() if smooth else sol.x_filt_mean())[:, k])
    # before t0: NaN record (the reference throws "Invalid t<t0")
    qm2, _ = sol(np.array([-0.1]))
    assert np.all(np.isnan(qm2))


def test_dense_output_adaptive_common_times(pkg):
    """BASELINE config 5 at test size: adaptive PI + RTS, posterior compared at saveat = 0:2^-4:t1 with the
    solver tolerance (as test/correctness.jl:62-66 compares the reference's dense output)."""
    vf = orc.vector_field("lorenz63")
    N, t1 = 66, 1.0
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-9, adaptive=True, max_steps=512)
    assert sol.retcode == ["Success"] * N
    tq = np.arange(0.0, t1 + 1e-12, 2.0**-4)
    qm, _ = sol(tq)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, 3)
    from scipy.integrate import solve_ivp

    for i in (0, 65):
        ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-9, adaptive=True)
        dense = np.array([orc.dense_output(ref, consts, float(t)).mu[:3] for t in tq])
        np.testing.assert_allclose(qm[i, :, :3], dense, rtol=1e-6, atol=1e-8)
        truth = solve_ivp(lambda t, u: np.array(vf.f(list(u), vf.p, t)), (0.0, t1), u0s[i], method="DOP853", rtol=1e-12,
                          atol=1e-12, t_eval=tq).y.T
        assert np.linalg.norm(qm[i, :, :3] - truth) <= 1e-3 * np.linalg.norm(truth)


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("q", [2, 5])
def test_dense_output_pleiades(pkg, q, smooth):
    """sol(t) on the workgroup-per-trajectory path (state dimension 84 / 168, csrc/dense_mfma.h) against the oracle's
    dense output (src/solution.jl:165-210): inside the grid, at a stored time, beyond the last time, before t0.
    Order 5 runs with dt = 2^-6: at config 4's 2^-10 the first residuals of an order-5 solve are h^5-small and every
    covariance (proportional to the diffusion estimates) is rounding noise in the reference arithmetic itself
    (tests/golden/make_exact.py); at 2^-6 the oracle is within 3e-7 of the extended-precision covariance, so the
    interpolated covariances ARE compared at order 5 too."""
    vf = orc.vector_field("pleiades")
    N, ns, dt = 3, 12, (2.0**-10 if q < 5 else 2.0**-6)
    ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, ns * dt), ()), perturb_scale=1e-3, n_perturbed=14)
    sol = pkg.solve(ens, pkg.EK1(order=q, smooth=smooth), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    grid = sol.t
    tq = np.array([0.3 * dt, 4.5 * dt, grid[7], 11.25 * dt, ns * dt + 0.4 * dt])
    qm, qc = sol(tq)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-3, n_perturbed=14)
    consts = orc.make_consts(28, q)
    for i in (0, 2):
        ref = orc.solve(vf, orc.EK1(order=q, smooth=smooth), u0=u0s[i], tspan=(0.0, ns * dt), dt=dt)
        for j, t in enumerate(tq):
            r = orc.dense_output(ref, consts, float(t), smoothed=smooth)
            np.testing.assert_allclose(qm[i, j, :28], r.mu[:28], rtol=1e-10, atol=1e-13, err_msg=f"traj {i} t={t}")
            c = r.cov()
            # on the scale of its largest entry: 1e-6 at order 2; 1e-4 at order 5 (oracle - extended precision there: 3e-7 in the
            # filter covariance, 3e-9 in the smoothed one; the Joseph-form kernels sit within 16x of that, test_pleiades_ensemble_parity)
            assert np.abs(qc[i, j] - c).max() <= (1e-6 if q < 5 else 1e-4) * np.abs(c).max() + 1e-300, (i, j)
            assert (np.diag(qc[i, j]) >= 0).all()
    np.testing.assert_array_equal(qm[:, 2], (sol.x_smooth_mean() if smooth else sol.x_filt_mean())[:, 7])
    qm2, _ = sol(np.array([-0.1]))
    assert np.all(np.isnan(qm2))


def test_dense_output_requires_smoothing(pkg):
    vf = orc.vector_field("lorenz63")
    ctx = pkg.Context("lorenz63", 3, 1, 64, smooth=True)
    ctx.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
    ctx.solve_fixed(np.arange(9) * 2.0**-9)
    with pytest.raises(pkg.OdefError, match="odef_smooth has not run"):
        ctx.dense_output([0.001], True)
    ctx.close()


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("q", [4, 5])
def test_dense_output_row_teams(pkg, q, smooth):
    """sol(t) for 12 < state dimension <= 32 (Lorenz-63 at orders 4 and 5: D = 15 on teams of 16 lanes, D = 18 on teams of 32;
    csrc/dense_rows.h) against the oracle's dense output: inside the grid, at a stored time, beyond the last time, before t0."""
    vf = orc.vector_field("lorenz63")
    N, dt, t1 = 9, 2.0**-6, 0.25
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=q, smooth=smooth), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    grid = sol.t
    tq = np.concatenate([np.linspace(0.0, t1, 9) + 0.3 * dt, [grid[3], t1 + 0.01]])
    qm, qc = sol(tq)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, q)
    for i in (0, 8):
        ref = orc.solve(vf, orc.EK1(order=q, smooth=smooth), u0=u0s[i], tspan=(0.0, t1), dt=dt)
        for j, t in enumerate(tq):
            r = orc.dense_output(ref, consts, float(t), smoothed=smooth)
            np.testing.assert_allclose(qm[i, j, :3], r.mu[:3], rtol=1e-9, atol=1e-12, err_msg=f"traj {i} t={t}")
            c = r.cov()
            assert np.abs(qc[i, j] - c).max() <= 1e-5 * np.abs(c).max() + 1e-300, (i, j)
    np.testing.assert_array_equal(qm[:, 9], (sol.x_smooth_mean() if smooth else sol.x_filt_mean())[:, 3])
    qm2, _ = sol(np.array([-0.1]))
    assert np.all(np.isnan(qm2))


# ---- posterior sampling -----------------------------------------------------------------------------------


@pytest.mark.parametrize("adaptive", [False, True])
def test_posterior_sampling(pkg, adaptive):
    """sample / sample_states (src/solution_sampling.jl) against the oracle with the same noise stream and square root,
    plus the reference's own acceptance test (test/solution.jl:57-72: < 5 % outside 3 sigma)."""
    vf = orc.vector_field("lorenz63")
    N, t1, n, seed = 70, 0.5, 10, 99
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    if adaptive:
        sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-9, adaptive=True, max_steps=256)
    else:
        sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-6, adaptive=False)
    st = sol.sample_states(n, seed)
    smp = sol.sample(n, seed)
    cap = st.shape[1]
    assert st.shape == (N, cap, 12, n) and smp.shape == (N, cap, 3, n)
    np.testing.assert_array_equal(smp, st[:, :, :3, :])
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, 3)
    for i in (0, 69):
        if adaptive:
            ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-9, adaptive=True)
        else:
            ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-6)
        ns = len(ref.t)
        raw = sol.raw_index(i)  # device slot of the k-th accepted record (adaptive solves keep rejected attempts too)
        assert len(raw) == ns
        raw = np.append(raw[1:] - 1, raw[-1])  # the draw of a repeated (rejected-attempt) state is made at its LAST repeat
        want = orc.sample_states(ref, consts, n, sqrt="cholesky",
                                 normal=lambda j, slot, k: orc.sample_normal(seed, i, j, int(raw[slot]), k, n, cap, 12))
        scale = np.abs(want).max(axis=(0, 2))[None, :, None]
        err = (np.abs(st[i, :ns] - want) / scale).max(axis=(0, 2))
        assert err[:3].max() < (1e-6 if adaptive else 1e-8) and err.max() < 1e-3, err
        # the reference's acceptance test on the device samples
        x = ref.means(smoothed=True)
        stds = np.sqrt(np.array([np.diag(c) for c in ref.covs(smoothed=True)]))
        out = np.abs(x[1:, :, None] - st[i, 1:ns]) > 3 * stds[1:, :, None] + 1e-12 * np.abs(x[1:, :, None])
        assert out.sum() < 0.05 * out.size
    # zero noise: the chain of conditional means is the smoothed mean
    z = sol.sample_states(1, seed, noise_scale=0.0)[..., 0]
    sm = sol.x_smooth_mean()
    k = 5
    np.testing.assert_allclose(z[:, 1:k, :3], sm[:, 1:k, :3], rtol=1e-7)
    # reproducible, and a different seed gives different draws
    np.testing.assert_array_equal(sol.sample_states(n, seed), st)
    assert not np.array_equal(sol.sample_states(n, seed + 1), st)


@pytest.mark.parametrize("q", [4, 5])
def test_posterior_sampling_row_teams(pkg, q):
    """sample_states for 12 < state dimension <= 32 (Lorenz-63 at orders 4 / 5, D = 15 / 18: csrc/sample_rows.h) against the oracle
    with the same noise stream and square root, the reference's acceptance test (test/solution.jl:57-72), the zero-noise chain and
    reproducibility -- as test_posterior_sampling does for D = 12."""
    vf = orc.vector_field("lorenz63")
    D = 3 * (q + 1)
    N, t1, n, seed = 11, 0.25, 8, 4242
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=q), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-6, adaptive=False)
    st = sol.sample_states(n, seed)
    cap = st.shape[1]
    assert st.shape == (N, cap, D, n) and np.isfinite(st).all()
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, q)
    for i in (0, 10):
        ref = orc.solve(vf, orc.EK1(order=q), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-6)
        ns = len(ref.t)
        want = orc.sample_states(ref, consts, n, sqrt="cholesky", normal=lambda j, slot, k: orc.sample_normal(seed, i, j, slot, k, n, cap, D))
        scale = np.abs(want).max(axis=(0, 2))[None, :, None]
        err = (np.abs(st[i, :ns] - want) / scale).max(axis=(0, 2))
        assert err[:3].max() < 1e-7 and err.max() < 1e-2, err
        x = ref.means(smoothed=True)
        stds = np.sqrt(np.array([np.diag(c) for c in ref.covs(smoothed=True)]))
        out = np.abs(x[1:, :, None] - st[i, 1:ns]) > 3 * stds[1:, :, None] + 1e-12 * np.abs(x[1:, :, None])
        assert out.sum() < 0.05 * out.size
    z = sol.sample_states(1, seed, noise_scale=0.0)[..., 0]
    np.testing.assert_allclose(z[:, 1:5, :3], sol.x_smooth_mean()[:, 1:5, :3], rtol=1e-7)
    np.testing.assert_array_equal(sol.sample_states(n, seed), st)
    assert not np.array_equal(sol.sample_states(n, seed + 1), st)
    # dense grid (src/solution_sampling.jl:63-75): shapes, finiteness, and the first block stays within 6 sigma of the smoothed mean
    tq = np.linspace(0.0, t1, 13)
    ds, _ = sol.dense_sample_states(4, seed, times=tq)
    assert ds.shape == (N, len(tq), D, 4) and np.isfinite(ds).all()


@pytest.mark.parametrize("q", [2, 5])
def test_posterior_sampling_pleiades(pkg, q):
    """sample_states on the workgroup-per-trajectory path (state dimension 84 / 168, csrc/sample_mfma.h): at order 2 value by value
    against the oracle with the same noise stream and square root; at both orders the zero-noise chain of conditional means is
    the smoothed mean, draws are finite and reproducible, and the dense-grid mode runs."""
    vf = orc.vector_field("pleiades")
    D = 28 * (q + 1)
    N, ns_, dt, n, seed = 2, 8, 2.0**-10, 3, 777
    ens = pkg.EnsembleProblem(pkg.ODEProblem("pleiades", vf.u0, (0.0, ns_ * dt), ()), perturb_scale=1e-3, n_perturbed=14)
    sol = pkg.solve(ens, pkg.EK1(order=q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    st = sol.sample_states(n, seed)
    cap = st.shape[1]
    assert st.shape == (N, cap, D, n) and np.isfinite(st).all()
    z = sol.sample_states(1, seed, noise_scale=0.0)[..., 0]
    np.testing.assert_allclose(z[:, 1:, :28], sol.x_smooth_mean()[:, 1:, :28], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(sol.sample_states(n, seed), st)
    assert not np.array_equal(sol.sample_states(n, seed + 1), st)
    if q == 2:
        u0s = orc.ensemble_u0(vf.u0, N, 1e-3, n_perturbed=14)
        consts = orc.make_consts(28, q)
        for i in (0, 1):
            ref = orc.solve(vf, orc.EK1(order=q), u0=u0s[i], tspan=(0.0, ns_ * dt), dt=dt)
            want = orc.sample_states(ref, consts, n, sqrt="cholesky", normal=lambda j, slot, k: orc.sample_normal(seed, i, j, slot, k, n, cap, D))
            scale = np.abs(want).max(axis=(0, 2))[None, :, None]
            err = (np.abs(st[i, :len(ref.t)] - want) / scale).max(axis=(0, 2))
            assert err[:28].max() < 1e-7 and err.max() < 1e-2, err
    tq = np.linspace(0.0, ns_ * dt, 7)
    ds, _ = sol.dense_sample_states(2, seed, times=tq)
    assert ds.shape == (N, len(tq), D, 2) and np.isfinite(ds).all()


@pytest.mark.parametrize("adaptive", [False, True])
def test_dense_posterior_sampling(pkg, adaptive):
    """dense_sample / dense_sample_states (src/solution_sampling.jl:63-75) against the oracle with the same noise stream
    and square root; shapes as test/solution.jl:74-79, 98-103 (1 000 times by default)."""
    vf = orc.vector_field("lorenz63")
    N, t1, n, seed = 70, 0.5, 4, 7
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, t1), vf.p), perturb_scale=1e-2)
    if adaptive:
        sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-9, adaptive=True, max_steps=256)
    else:
        sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, dt=2.0**-6, adaptive=False)
    ds, times = sol.dense_sample_states(n, seed)
    du, times2 = sol.dense_sample(n, seed)
    assert ds.shape == (N, 1000, 12, n) and du.shape == (N, 1000, 3, n) and len(times) == 1000
    assert times[0] == 0.0 and times[-1] == t1
    np.testing.assert_array_equal(du, ds[:, :, :3, :])
    tq = np.linspace(0.0, t1, 57)
    st, _ = sol.dense_sample_states(n, seed, times=tq)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, 3)
    for i in (0, 69):
        if adaptive:
            ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-9, adaptive=True)
        else:
            ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, t1), dt=2.0**-6)
        want, _ = orc.dense_sample_states(ref, consts, n, times=tq, sqrt="cholesky", seed=seed, traj=i)
        scale = np.abs(want).max(axis=(0, 2))[None, :, None]
        err = (np.abs(st[i] - want) / scale).max(axis=(0, 2))
        assert err[:3].max() < (1e-6 if adaptive else 1e-8) and err.max() < 1e-3, err


@pytest.mark.parametrize("adaptive", [False, True])
def test_distributed_solve_single_rank(pkg, adaptive, monkeypatch):
    """EnsembleHIP(distributed=True) with one rank (the N > 1 exchange is covered by the gloo tests): the shard is the whole
    ensemble, gather_final() is the final posterior mean of every trajectory (the last kept record of an adaptive solve)."""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    vf = orc.vector_field("lorenz63")
    N = 130
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, 0.25), vf.p), perturb_scale=1e-2)
    kw = dict(adaptive=True, dt=2.0**-9, max_steps=256) if adaptive else dict(adaptive=False, dt=2.0**-7)
    sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(distributed=True), trajectories=N, **kw)
    assert sol.shard == (0, N, N, 1)
    fin = sol.gather_final()
    assert fin.shape == (N, 12)
    xf = sol.x_filt_mean()
    last = np.asarray(sol.nsaved) - 1 if adaptive else np.full(N, xf.shape[1] - 1)
    np.testing.assert_array_equal(fin, xf[np.arange(N), last])
    ref = orc.solve(vf, orc.EK1(order=3), u0=orc.ensemble_u0(vf.u0, N, 1e-2)[129], tspan=(0.0, 0.25),
                    **(dict(adaptive=True, dt=2.0**-9) if adaptive else dict(dt=2.0**-7)))
    np.testing.assert_allclose(fin[129, :3], ref.means(smoothed=False)[-1, :3], rtol=1e-6 if adaptive else 1e-10)


def test_sampling_needs_smoothing_solution(pkg):
    vf = orc.vector_field("lorenz63")
    prob = pkg.ODEProblem("lorenz63", vf.u0, (0.0, 0.1), vf.p)
    sol = pkg.solve(prob, pkg.EK1(order=3, smooth=False), dt=2.0**-6, adaptive=False)
    with pytest.raises(AssertionError, match="non-smoothed"):
        sol.sample(2)


# ---- run-time compiled vector fields ------------------------------------------------------------------------

USER_LORENZ = """
struct UserLorenz {
  static constexpr int d = 3, np = 3;
  template <class T>
  __device__ static void f(const T (&u)[3], const double* p, T (&du)[3]) {
    const double s = p[0], r = p[1], b = p[2];
    du[0] = s * (u[1] - u[0]);
    du[1] = u[0] * (r - u[2]) - u[1];
    du[2] = u[0] * u[1] - b * u[2];
  }
  __device__ static void jac(const double (&u)[3], const double* p, double (&J)[3][3]) {
    const double s = p[0], r = p[1], b = p[2];
    J[0][0] = -s;       J[0][1] = s;    J[0][2] = 0.0;
    J[1][0] = r - u[2]; J[1][1] = -1.0; J[1][2] = -u[0];
    J[2][0] = u[1];     J[2][1] = u[0]; J[2][2] = -b;
  }
};
"""

# Lorenz-96 with four variables: du_i = (u_{i+1} - u_{i-2}) u_{i-1} - u_i + F  (d = 4: no compiled-in kernel has it)
USER_L96 = """
struct UserL96 {
  static constexpr int d = 4, np = 1;
  template <class T>
  __device__ static void f(const T (&u)[4], const double* p, T (&du)[4]) {
    du[0] = (u[1] - u[2]) * u[3] - u[0] + p[0];
    du[1] = (u[2] - u[3]) * u[0] - u[1] + p[0];
    du[2] = (u[3] - u[0]) * u[1] - u[2] + p[0];
    du[3] = (u[0] - u[1]) * u[2] - u[3] + p[0];
  }
  __device__ static void jac(const double (&u)[4], const double* p, double (&J)[4][4]) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) J[i][j] = 0.0;
    for (int i = 0; i < 4; ++i) {
      const int ip = (i + 1) % 4, im1 = (i + 3) % 4, im2 = (i + 2) % 4;
      J[i][ip] += u[im1];
      J[i][im2] -= u[im1];
      J[i][im1] += u[ip] - u[im2];
      J[i][i] -= 1.0;
    }
  }
};
"""


def _l96_field():
    def f(u, p, t):
        return [(u[(i + 1) % 4] - u[(i + 2) % 4]) * u[(i + 3) % 4] - u[i] + p[0] for i in range(4)]

    def jac(u, p, t):
        J = np.zeros((4, 4))
        for i in range(4):
            ip, im1, im2 = (i + 1) % 4, (i + 3) % 4, (i + 2) % 4
            J[i, ip] += u[im1]
            J[i, im2] -= u[im1]
            J[i, im1] += u[ip] - u[im2]
            J[i, i] -= 1.0
        return J

    return orc.VectorField("l96", 100, 4, 1, f, jac, np.array([1.0, 2.0, 0.5, -1.0]), np.array([8.0]), (0.0, 0.25))


@pytest.mark.parametrize("kernels", ["lane", "rows"])
def test_user_vector_field_equals_the_compiled_in_one(pkg, kernels, monkeypatch):
    """The same Lorenz-63 text through odef_rhs_compile (hipcc child process) and through the compiled-in registry: identical
    kernels source, so identical results -- fixed grid + smoother, adaptive, dense output, sampling.  Run-time compiled
    fields get BOTH kernel families of the small state dimensions (round 3): one lane per trajectory, and -- below the same
    ensemble sizes as the compiled-in fields -- 16 lanes per trajectory (rows_kernels.h); each is compared with the
    compiled-in kernels of its own family, pinned by the launcher's environment switches."""
    big = "1000000000"
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", "0" if kernels == "lane" else big)
    monkeypatch.setenv("ODEF_SMOOTH_ROWS_MAX_N", "0" if kernels == "lane" else big)
    monkeypatch.setenv("ODEF_SMOOTH_LANE_MIN_N", "1")  # (70 trajectories would otherwise go to the LDS row-team smoother)
    pkg.compile_rhs("UserLorenz", USER_LORENZ, 3, 3)
    vf = orc.vector_field("lorenz63")
    N = 70
    for adaptive in (False, True):
        sols = []
        for rhs in ("lorenz63", "UserLorenz"):
            ens = pkg.EnsembleProblem(pkg.ODEProblem(rhs, vf.u0, (0.0, 0.25), vf.p), perturb_scale=1e-2)
            kw = dict(dt=2.0**-9, adaptive=True, max_steps=256) if adaptive else dict(dt=2.0**-7, adaptive=False)
            sols.append(pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, **kw))
        a, b = sols
        assert b.retcode == ["Success"] * N
        want = {("lane", False): ("odef_jit_fixed_every", "odef_jit_smooth_fixed"), ("lane", True): ("odef_jit_adaptive", "odef_jit_smooth_adapt"),
                ("rows", False): ("odef_jit_rows_fixed_every", "odef_jit_bcast_fixed"), ("rows", True): ("odef_jit_rows_adaptive", "odef_jit_bcast_adapt")}
        assert (b.ctx.kernel_name(0), b.ctx.kernel_name(1)) == want[kernels, adaptive]
        assert ("rows" in a.ctx.kernel_name(0)) == (kernels == "rows") and ("bcast" in a.ctx.kernel_name(1)) == (kernels == "rows")
        # same source, same compiler back end; tolerances only allow for a different contraction/scheduling choice
        # (higher-derivative components amplify one ulp, tests/_parity.py)
        np.testing.assert_allclose(b.x_filt_mean()[..., :3], a.x_filt_mean()[..., :3], rtol=1e-12, atol=0)
        np.testing.assert_allclose(b.x_smooth_mean()[..., :3], a.x_smooth_mean()[..., :3], rtol=1e-11, atol=0)
        assert P.cov_err(b.x_smooth_cov()[0], a.x_smooth_cov()[0]) < 1e-6
        np.testing.assert_array_equal(b.destats.naccept, a.destats.naccept)
        tq = np.linspace(0.0, 0.25, 7)
        np.testing.assert_allclose(b(tq)[0][..., :3], a(tq)[0][..., :3], rtol=1e-11, atol=0)
        sa, sb = a.sample_states(3, 7)[:, :, :3], b.sample_states(3, 7)[:, :, :3]
        assert np.abs(sb - sa).max() <= 1e-8 * np.abs(sa).max()


def _l96_source(name, d):
    return f"""
struct {name} {{
  static constexpr int d = {d}, np = 1;
  template <class T>
  __device__ static void f(const T (&u)[{d}], const double* p, T (&du)[{d}]) {{
    for (int i = 0; i < {d}; ++i) du[i] = (u[(i + 1) % {d}] - u[(i + {d - 2}) % {d}]) * u[(i + {d - 1}) % {d}] - u[i] + p[0];
  }}
}};
"""


@pytest.mark.parametrize("d,q", [(5, 2), (8, 1)])
def test_user_vector_field_larger_state(pkg, d, q):
    """Lorenz-96 with d variables, D = d(q+1) = 15 / 16: the lane filter plus the row-team smoother and dense output, all
    run-time compiled; EK1 with the forward-mode Jacobian."""
    name = f"UserL96d{d}"
    pkg.compile_rhs(name, _l96_source(name, d), d, 1)

    def f(u, p, t):
        return [(u[(i + 1) % d] - u[(i + d - 2) % d]) * u[(i + d - 1) % d] - u[i] + p[0] for i in range(d)]

    def jac(u, p, t):
        J = np.zeros((d, d))
        for i in range(d):
            ip, im2, im1 = (i + 1) % d, (i + d - 2) % d, (i + d - 1) % d
            J[i, ip] += u[im1]
            J[i, im2] -= u[im1]
            J[i, im1] += u[ip] - u[im2]
            J[i, i] -= 1.0
        return J

    u0 = np.array([1.0, 2.0, 0.5, -1.0, 0.3, 1.5, -0.7, 0.9])[:d]
    vf = orc.VectorField(name, 100, d, 1, f, jac, u0, np.array([8.0]), (0.0, 0.2))
    prob = pkg.ODEProblem(name, vf.u0, vf.tspan, vf.p)
    sol = pkg.solve(prob, pkg.EK1(order=q), dt=2.0**-7, adaptive=False)
    assert sol.retcode == ["Success"]
    ref = orc.solve(vf, orc.EK1(order=q), tspan=vf.tspan, dt=2.0**-7)
    np.testing.assert_allclose(sol.x_filt_mean()[0][:, :d], ref.means(smoothed=False)[:, :d], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(sol.u[0], ref.u, rtol=1e-9, atol=1e-13)
    assert P.cov_err(sol.x_smooth_cov()[0], ref.covs(smoothed=True)) < 1e-5
    # sol(t) on the run-time compiled row-team dense kernel (csrc/dense_rows.h), smoothed posterior
    tq = np.array([0.013, 0.1, 2.0**-7 * 5, 0.2 + 0.004])
    qm, qc = sol(tq)
    consts = orc.make_consts(d, q)
    for j, t in enumerate(tq):
        r = orc.dense_output(ref, consts, float(t), smoothed=True)
        np.testing.assert_allclose(qm[0, j, :d], r.mu[:d], rtol=1e-9, atol=1e-12)
        assert np.abs(qc[0, j] - r.cov()).max() <= 1e-5 * np.abs(r.cov()).max() + 1e-300
    # ... and posterior sampling on the run-time compiled row teams (csrc/sample_rows.h): the zero-noise chain of conditional
    # means is the smoothed mean, draws are finite and reproducible
    zc = sol.sample_states(1, 7, noise_scale=0.0)[..., 0]
    np.testing.assert_allclose(zc[0, 1:6, :d], sol.x_smooth_mean()[0, 1:6, :d], rtol=1e-7, atol=1e-10)
    s1 = sol.sample_states(3, 7)
    assert np.isfinite(s1).all()
    np.testing.assert_array_equal(sol.sample_states(3, 7), s1)


def test_user_vector_field_on_the_matrix_core_kernels(pkg):
    """A user vector field ABOVE state dimension 20: Lorenz-96 with 12 variables at order 2 (D = 36; no compiled-in field has
    d = 12, and the struct has no `jac`).  odef_rhs_compile builds the workgroup-per-trajectory kernels of csrc/filter_mfma.h /
    smooth_mfma.h / dense_mfma.h around it for this one order and algorithm -- with their host-side launch code, as a shared
    object that exports the launch table (csrc/jit.hip, team_translation_unit) -- and the context runs on it like on a
    compiled-in field: fixed grid filter (forward-mode Jacobian on one lane of the helper wavefront) + split-pass smoother
    against the oracle at the oracle's rounding-noise level, the adaptive filter against the oracle's controller loop, sol(t)."""
    d, q, name = 12, 2, "UserL96d12"
    pkg.compile_rhs(name, _l96_source(name, d), d, 1)

    def f(u, p, t):
        return [(u[(i + 1) % d] - u[(i + d - 2) % d]) * u[(i + d - 1) % d] - u[i] + p[0] for i in range(d)]

    def jac(u, p, t):
        J = np.zeros((d, d))
        for i in range(d):
            ip, im2, im1 = (i + 1) % d, (i + d - 2) % d, (i + d - 1) % d
            J[i, ip] += u[im1]
            J[i, im2] -= u[im1]
            J[i, im1] += u[ip] - u[im2]
            J[i, i] -= 1.0
        return J

    u0 = np.array([1.0, 2.0, 0.5, -1.0, 0.3, 1.5, -0.7, 0.9, 1.2, -0.4, 0.8, 2.1])
    vf = orc.VectorField(name, 100, d, 1, f, jac, u0, np.array([8.0]), (0.0, 0.1))
    N, ns, dt = 3, 12, 2.0**-7
    ens = pkg.EnsembleProblem(pkg.ODEProblem(name, vf.u0, (0.0, ns * dt), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] * N
    assert f"ek_filter_mfma_kernel<odef::{name}" in sol.ctx.kernel_name(0) and "rts_smooth_sweeps_kernel<12" in sol.ctx.kernel_name(1)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    alg_o = orc.Alg("EK1", q, "dynamic", True)
    mf, ms, cf, cs = sol.x_filt_mean(), sol.x_smooth_mean(), sol.x_filt_cov(), sol.x_smooth_cov()
    for i in (0, 2):
        for smoothed, m, c in ((False, mf, cf), (True, ms, cs)):
            base, nm, nc = P.oracle_noise(vf, alg_o, u0s[i], dict(tspan=(0.0, ns * dt), dt=dt), smoothed)
            P.check_against_oracle(m[i], c[i], base.means(smoothed=smoothed), base.covs(smoothed=smoothed), d, nm, nc,
                                   f"user L96 d=12 traj {i} smoothed={smoothed}")
    # sol(t) (csrc/dense_mfma.h), smoothed posterior
    consts = orc.make_consts(d, q)
    ref = orc.solve(vf, orc.EK1(order=q), u0=u0s[0], tspan=(0.0, ns * dt), dt=dt)
    tq = np.array([0.013, 2.0**-7 * 5, ns * dt])
    qm, _ = sol(tq)
    want = np.array([orc.dense_output(ref, consts, float(t), smoothed=True).mu[:d] for t in tq])
    np.testing.assert_allclose(qm[0][:, :d], want, rtol=1e-8, atol=1e-11)
    # the adaptive filter of the same module
    sol = pkg.solve(pkg.ODEProblem(name, vf.u0, vf.tspan, vf.p), pkg.EK1(order=q, smooth=False), dt=2.0**-8, adaptive=True, abstol=1e-7, reltol=1e-5,
                    max_steps=256)
    assert sol.retcode == ["Success"]
    ref = orc.solve(vf, orc.Alg("EK1", q, "dynamic", False), tspan=vf.tspan, dt=2.0**-8, adaptive=True, abstol=1e-7, reltol=1e-5)
    n = len(ref.t)
    assert int(sol.nsaved[0]) == n and int(sol.destats.nreject[0]) == ref.nreject
    np.testing.assert_allclose(sol.t[0][:n], ref.t, rtol=1e-6)
    np.testing.assert_allclose(sol.u[0][:n], ref.u, rtol=1e-6, atol=1e-9)
    # what no kernel covers is refused with a reason: odd d above state dimension 20
    pkg.compile_rhs("UserL96d7", _l96_source("UserL96d7", 7), 7, 1)
    with pytest.raises(pkg.OdefError, match="even d"):
        pkg.Context("UserL96d7", 3, 1, 2)


@pytest.mark.parametrize("d,q", [(12, 2), (8, 3), (4, 5), (14, 2)])
def test_matrix_core_filter_reads_no_lds_it_did_not_write(pkg, d, q, monkeypatch):
    """Shapes with fewer than 16 rows per derivative block run the workgroup-per-trajectory filter with partly filled 16 x 16
    blocks.  A diagnostic build ($ODEFILTER_HIP_JIT_FLAGS -> ODEF_MF_DEBUG_FILL, csrc/filter_mfma.h) starts every workgroup
    with its LDS full of NaNs: whatever the kernel reads without having written it poisons the result.  (Found this way: the
    factorisation of H Q H' copied a 16 x 16 block out of a d x d matrix without looking at d -- results depended on what the
    previous kernel had left in LDS.)  d = 14 is not a multiple of 4: its smoother takes Y' = A X through the workspace, the others
    form it on chip."""
    monkeypatch.setenv("ODEFILTER_HIP_JIT_FLAGS", "-DODEF_MF_DEBUG_FILL=1 -DODEF_MF_DEBUG_LO=0 -DODEF_MF_DEBUG_HI=W::size")
    name = f"PoisonL96d{d}q{q}"
    pkg.compile_rhs(name, _l96_source(name, d), d, 1)

    def f(u, p, t):
        return [(u[(i + 1) % d] - u[(i + d - 2) % d]) * u[(i + d - 1) % d] - u[i] + p[0] for i in range(d)]

    def jac(u, p, t):
        J = np.zeros((d, d))
        for i in range(d):
            ip, im2, im1 = (i + 1) % d, (i + d - 2) % d, (i + d - 1) % d
            J[i, ip] += u[im1]
            J[i, im2] -= u[im1]
            J[i, im1] += u[ip] - u[im2]
            J[i, i] -= 1.0
        return J

    u0 = 1.0 + np.random.default_rng(d).normal(size=d)
    vf = orc.VectorField(name, 100, d, 1, f, jac, u0, np.array([8.0]), (0.0, 0.1))
    ns, dt = 12, 2.0**-7
    sol = pkg.solve(pkg.ODEProblem(name, vf.u0, (0.0, ns * dt), vf.p), pkg.EK1(order=q), dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] and "ek_filter_mfma_kernel" in sol.ctx.kernel_name(0)
    ref = orc.solve(vf, orc.EK1(order=q), tspan=(0.0, ns * dt), dt=dt)
    np.testing.assert_allclose(sol.u[0], ref.u, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sol.x_filt_mean()[0][:, :d], ref.means(smoothed=False)[:, :d], rtol=1e-10, atol=1e-13)
    assert np.isfinite(sol.x_smooth_cov()[0]).all() and np.isfinite(sol.x_filt_cov()[0]).all()
    assert P.cov_err(sol.x_smooth_cov()[0], ref.covs(smoothed=True)) < 1e-5


def test_user_vector_field_on_the_row_team_kernels_at_config2_size(pkg):
    """A user vector field with d = 5 at order 2 (D = 15: no compiled-in kernel has this shape) and 4 096 trajectories -- the
    ensemble size of BASELINE config 2, where the lane kernels would occupy 64 of the chip's 1 024 SIMDs: the library must
    pick the run-time compiled 16-lanes-per-trajectory filter and smoother (asked through odef_kernel_name, not assumed), and
    their results must agree with the oracle and with the lane filter + LDS row-team smoother forced on the same ensemble."""
    name, d, q = "UserL96d5cfg2", 5, 2
    pkg.compile_rhs(name, _l96_source(name, d), d, 1)

    def f(u, p, t):
        return [(u[(i + 1) % d] - u[(i + d - 2) % d]) * u[(i + d - 1) % d] - u[i] + p[0] for i in range(d)]

    def jac(u, p, t):
        J = np.zeros((d, d))
        for i in range(d):
            ip, im2, im1 = (i + 1) % d, (i + d - 2) % d, (i + d - 1) % d
            J[i, ip] += u[im1]
            J[i, im2] -= u[im1]
            J[i, im1] += u[ip] - u[im2]
            J[i, i] -= 1.0
        return J

    u0 = np.array([1.0, 2.0, 0.5, -1.0, 0.3])
    vf = orc.VectorField(name, 100, d, 1, f, jac, u0, np.array([8.0]), (0.0, 0.25))
    N, dt = 4096, 2.0**-7
    ens = pkg.EnsembleProblem(pkg.ODEProblem(name, vf.u0, vf.tspan, vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] * N
    assert sol.ctx.kernel_name(0) == "odef_jit_rows_fixed_every" and sol.ctx.kernel_name(1) == "odef_jit_bcast_fixed"
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    mf, ms, cs = sol.x_filt_mean(), sol.x_smooth_mean(), sol.x_smooth_cov()
    for i in (0, 17, 4095):
        ref = orc.solve(vf, orc.EK1(order=q), u0=u0s[i], tspan=vf.tspan, dt=dt)
        np.testing.assert_allclose(mf[i][:, :d], ref.means(smoothed=False)[:, :d], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(ms[i][:, :d], ref.means(smoothed=True)[:, :d], rtol=1e-10, atol=1e-13)
        assert P.cov_err(cs[i], ref.covs(smoothed=True)) < 1e-5
    import os as _os

    old = {k: _os.environ.get(k) for k in ("ODEF_FILTER_ROWS_MAX_N", "ODEF_SMOOTH_ROWS_MAX_N")}
    try:
        _os.environ["ODEF_FILTER_ROWS_MAX_N"] = "0"
        _os.environ["ODEF_SMOOTH_ROWS_MAX_N"] = "0"
        lane = pkg.solve(ens, pkg.EK1(order=q), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
        assert lane.ctx.kernel_name(0) == "odef_jit_fixed_every" and lane.ctx.kernel_name(1) == "odef_jit_smooth_rows"
        np.testing.assert_allclose(ms[..., :d], lane.x_smooth_mean()[..., :d], rtol=1e-9, atol=1e-12)
    finally:
        for k, v in old.items():
            if v is None:
                _os.environ.pop(k, None)
            else:
                _os.environ[k] = v


def test_user_vector_field_without_jacobian_uses_forward_mode(pkg):
    """No `jac` in the struct: EK1 differentiates `f` with forward-mode duals, as the reference falls back to ForwardDiff
    (src/perform_step.jl:119-121).  Exact derivatives, so the result matches the analytic-Jacobian kernel."""
    src = USER_LORENZ[: USER_LORENZ.index("  __device__ static void jac")] + "};\n"
    pkg.compile_rhs("LorenzNoJac", src.replace("UserLorenz", "LorenzNoJac"), 3, 3)
    vf = orc.vector_field("lorenz63")
    sols = []
    for rhs in ("lorenz63", "LorenzNoJac"):
        prob = pkg.ODEProblem(rhs, vf.u0, (0.0, 0.5), vf.p)
        sols.append(pkg.solve(prob, pkg.EK1(order=3), dt=2.0**-7, adaptive=False))
    np.testing.assert_allclose(sols[1].u, sols[0].u, rtol=1e-11, atol=0)
    ref = orc.solve(vf, orc.EK1(order=3), tspan=(0.0, 0.5), dt=2.0**-7)
    np.testing.assert_allclose(sols[1].u[0], ref.u, rtol=1e-10, atol=0)


@pytest.mark.parametrize("kind", ["EK0", "EK1"])
def test_user_vector_field_new_dimension_against_oracle(pkg, kind):
    """A vector field no compiled-in kernel covers (d = 4, order 2, D = 12): filter + smoother + dense output."""
    pkg.compile_rhs("UserL96", USER_L96, 4, 1)
    vf = _l96_field()
    N, dt = 65, 2.0**-7
    ens = pkg.EnsembleProblem(pkg.ODEProblem("UserL96", vf.u0, vf.tspan, vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, _alg(pkg, kind, 2), pkg.EnsembleHIP(), trajectories=N, dt=dt, adaptive=False)
    assert sol.retcode == ["Success"] * N
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(4, 2)
    for i in (0, 64):
        ref = orc.solve(vf, orc.Alg(kind, 2, "dynamic", True), u0=u0s[i], tspan=vf.tspan, dt=dt)
        np.testing.assert_allclose(sol.x_filt_mean()[i][:, :4], ref.means(smoothed=False)[:, :4], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(sol.u[i], ref.u, rtol=1e-10, atol=1e-13)
        assert P.cov_err(sol.x_smooth_cov()[i], ref.covs(smoothed=True)) < 1e-6
        tq = np.array([0.01, 0.1, 0.2])
        qm, _ = sol(tq)
        want = np.array([orc.dense_output(ref, consts, float(t)).mu[:4] for t in tq])
        np.testing.assert_allclose(qm[i][:, :4], want, rtol=1e-9, atol=1e-12)


# ---- the reference's "specific problems" (test/specific_problems.jl) -------------------------------------------

USER_LOGISTIC = """
struct Logistic {
  static constexpr int d = 1, np = 1;
  template <class T>
  __device__ static void f(const T (&u)[1], const double* p, T (&du)[1]) { du[0] = p[0] * u[0] * (1.0 - u[0]); }
};
"""


@pytest.mark.parametrize("kind", ["EK0", "EK1"])
def test_one_dimensional_problem(pkg, kind):
    """test/specific_problems.jl:70-75 (OOP logistic problem, d = 1, order 4, default adaptive solve); EK1 differentiates
    f in forward mode.  A run-time compiled vector field: no compiled-in kernel has d = 1."""
    pkg.compile_rhs("Logistic", USER_LOGISTIC, 1, 1)
    f = lambda u, p, t: [p[0] * u[0] * (1.0 - u[0])]  # noqa: E731
    jac = lambda u, p, t: np.array([[p[0] * (1.0 - 2.0 * u[0])]])  # noqa: E731
    vf = orc.VectorField("logistic", 100, 1, 1, f, jac, np.array([0.1]), np.array([3.0]), (0.0, 5.0))
    prob = pkg.ODEProblem("Logistic", vf.u0, vf.tspan, vf.p)
    sol = pkg.solve(prob, _alg(pkg, kind, 4), adaptive=True, dt=5e-3, max_steps=1024)
    assert sol.retcode == ["Success"]
    ref = orc.solve(vf, orc.Alg(kind, 4, "dynamic", True), adaptive=True, dt=5e-3)
    n = int(sol.nsaved[0])
    assert n == len(ref.t)
    # the step sizes come out of a 5th root of an error estimate that is itself at rounding level relative to the state
    # (order 4 on a scalar logistic curve: the residual z is all cancellation): same accept/reject sequence, times equal
    # to ~1e-5 with the reference's exact reciprocals in inv(P), ~5e-5 with the kernel's division-free table
    np.testing.assert_allclose(sol.t[0, :n], ref.t, rtol=2e-4)
    np.testing.assert_allclose(sol.u[0, :n], ref.u, rtol=2e-4)
    exact = 0.1 * np.exp(15.0) / (1.0 + 0.1 * (np.exp(15.0) - 1.0))
    assert abs(sol.u[0, n - 1, 0] - exact) < 1e-4


def test_stiff_vanderpol(pkg):
    """test/specific_problems.jl:44-47: the stiff van der Pol problem runs through the adaptive EK1(3) (mu = 1e3 here, so
    that the oracle finishes in a second: 3 700 accepted steps)."""
    vf = orc.vector_field("vanderpol")
    p = np.array([1e3])
    prob = pkg.ODEProblem("vanderpol", vf.u0, (0.0, 6.3), p)
    sol = pkg.solve(prob, pkg.EK1(order=3, smooth=False), adaptive=True, dt=1e-3, max_steps=8192)
    assert sol.retcode == ["Success"]
    ref = orc.solve(vf, orc.EK1(order=3, smooth=False), p=p, tspan=(0.0, 6.3), adaptive=True, dt=1e-3)
    n = int(sol.nsaved[0])
    assert abs(n - len(ref.t)) <= 0.02 * len(ref.t)  # the accept/reject sequence of a stiff problem may differ by rounding
    assert sol.t[0, n - 1] == 6.3
    np.testing.assert_allclose(sol.u[0, n - 1], ref.u[-1], rtol=1e-3)


def test_non_uniform_fixed_grid(pkg):
    """`tstops` as the full grid with several distinct step sizes (one preconditioner table per distinct h, built on the
    host with the reference's running product, src/preconditioning.jl:9-14), filter + smoother + dense output."""
    vf = orc.vector_field("lorenz63")
    hs = np.array([2.0**-9, 2.0**-8, 3 * 2.0**-10, 2.0**-9, 2.0**-7, 2.0**-8, 5 * 2.0**-11] * 6)
    grid = np.concatenate([[0.0], np.cumsum(hs)])
    N = 66
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, float(grid[-1])), vf.p), perturb_scale=1e-2)
    sol = pkg.solve(ens, pkg.EK1(order=3), pkg.EnsembleHIP(), trajectories=N, adaptive=False, tstops=grid)
    np.testing.assert_array_equal(sol.t, grid)
    u0s = orc.ensemble_u0(vf.u0, N, 1e-2)
    consts = orc.make_consts(3, 3)
    for i in (0, 65):
        ref = orc.solve(vf, orc.EK1(order=3), u0=u0s[i], tspan=(0.0, float(grid[-1])), tgrid=grid)
        np.testing.assert_allclose(sol.x_filt_mean()[i][:, :3], ref.means(smoothed=False)[:, :3], rtol=1e-10)
        np.testing.assert_allclose(sol.u[i], ref.u, rtol=1e-9)
        assert P.cov_err(sol.x_smooth_cov()[i], ref.covs(smoothed=True)) < 1e-5
        tq = np.array([0.5 * (grid[3] + grid[4]), 0.9 * grid[-1]])
        want = np.array([orc.dense_output(ref, consts, float(t)).mu[:3] for t in tq])
        np.testing.assert_allclose(sol(tq)[0][i][:, :3], want, rtol=1e-8)


def test_random_configuration_sweep(pkg, monkeypatch):
    """tools/parity_sweep.py: 60 random (vector field, order, EK0/EK1, diffusion model, fixed/adaptive, u0, p, dt)
    configurations against the oracle; a configuration may only be waived when the oracle's own 1-ulp spread explains
    the difference, and a diverging one must be reported Unstable where the oracle raises."""
    import runpy

    monkeypatch.setattr(sys, "argv", ["parity_sweep.py", "5", "60"])
    with pytest.raises(SystemExit) as ex:
        runpy.run_path(os.path.join(os.path.dirname(GOLD), "..", "tools", "parity_sweep.py"), run_name="__main__")
    assert ex.value.code == 0


# ---- edge cases ------------------------------------------------------------------------------------------


def test_single_trajectory_and_single_step(pkg):
    """N = 1 (one lane of one wave), one step (n_t = 2), and `solve(ODEProblem, ...)` without an ensemble."""
    vf = orc.vector_field("fhn")
    prob = pkg.ODEProblem("fhn", vf.u0, (0.0, 7e-2), vf.p)
    sol = pkg.solve(prob, pkg.EK0(order=1), dt=7e-2, adaptive=False)
    ref = orc.solve(vf, orc.EK0(order=1), tspan=(0.0, 7e-2), dt=7e-2)
    assert sol.u.shape == (1, 2, 2) and len(sol.t) == 2
    np.testing.assert_allclose(sol.u[0], ref.u, rtol=1e-12)
    assert P.cov_err(sol.x_smooth_cov()[0], ref.covs(smoothed=True)) < 1e-9
    np.testing.assert_array_equal(sol.x_smooth_mean()[0, -1], sol.x_filt_mean()[0, -1])  # test/smoothing.jl:37


def test_config1_fhn_ek0_full(pkg):
    """BASELINE config 1 (examples/fitzhughnagumo_animation.jl:8-23): FHN, EK0(order=1), dt = 7e-2 on (0, 20)
    incl. the clipped last step -- through the GPU path, against the oracle."""
    vf = orc.vector_field("fhn")
    sol = pkg.solve(pkg.ODEProblem("fhn", vf.u0, vf.tspan, vf.p), pkg.EK0(order=1), dt=7e-2, adaptive=False)
    ref = orc.solve(vf, orc.EK0(order=1), dt=7e-2)
    assert len(sol.t) == 287 and sol.t[-1] == 20.0
    np.testing.assert_allclose(sol.u[0], ref.u, rtol=1e-10)
    np.testing.assert_allclose(sol.log_likelihood[0], ref.log_likelihood, rtol=1e-8)
    assert sol.destats.nf[0] == 286 and sol.destats.njacs[0] == 0 and sol.destats.naccept[0] == 286


def test_unstable_trajectory_does_not_abort_the_batch(pkg):
    """A diverging trajectory gets RETCODE Unstable; its neighbours are unaffected (the reference would throw)."""
    vf = orc.vector_field("lorenz63")
    u0s = np.tile(vf.u0, (4, 1))
    u0s[2] = [1e200, 1e200, 1e200]
    prob = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, 0.125), vf.p), u0s=u0s)
    sol = pkg.solve(prob, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=2.0**-6, adaptive=False)
    rc = sol.retcode
    assert rc[2] == "Unstable" and rc[0] == rc[1] == rc[3] == "Success"
    ref = orc.solve(vf, orc.EK1(order=3, smooth=False), tspan=(0.0, 0.125), dt=2.0**-6)
    np.testing.assert_allclose(sol.u[3], ref.u, rtol=1e-12)


def test_adaptive_max_steps_reports_maxiters(pkg):
    vf = orc.vector_field("lorenz63")
    prob = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", vf.u0, (0.0, 2.0), vf.p), u0s=np.tile(vf.u0, (3, 1)))
    sol = pkg.solve(prob, pkg.EK1(order=3, smooth=False), pkg.EnsembleHIP(), dt=2.0**-9, adaptive=True, max_steps=16)
    assert sol.retcode == ["MaxIters"] * 3 and np.all(sol.ctx.get(9) == 17) and np.all(sol.nsaved <= 17)


# ---- multi-GPU group of the C ABI (odef_group_*, odef_allgather) ---------------------------------------------------


def test_group_one_device_allgather_through_rccl(pkg):
    """The single-process multi-GPU entry points with ONE device: shard = whole ensemble, odef_allgather goes through
    ncclCommInitAll / ncclAllGather (librccl, world of one) and returns the final posterior means a plain context gives."""
    from odefilters_jl_amd import host

    vf = orc.vector_field("lorenz63")
    N, ns, dt = 1000, 48, 2.0**-9
    tg = np.arange(ns + 1) * dt
    with host.DeviceGroup("lorenz63", 3, 1, N, 1, smooth=True) as grp:
        assert grp.shard(0) == (0, N)
        grp.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
        grp.solve_fixed(tg)
        grp.smooth()
        fin = grp.allgather(smoothed=False)
        fin_s = grp.allgather(smoothed=True)
    ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
    ctx.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
    ctx.solve_fixed(tg)
    ctx.smooth()
    np.testing.assert_array_equal(fin, ctx.get(0)[-1])
    np.testing.assert_array_equal(fin_s, ctx.get(11)[-1])  # the last smoothed state is the last filter state
    ctx.close()


@pytest.mark.parametrize("adaptive", [False, True])
def test_group_two_shards_equal_one_solve(pkg, adaptive):
    """Two shards (both on this machine's one GPU: the gather falls back to device copies, the sharding, the concurrent
    launches and the packing of the final records are the multi-GPU code) against ONE solve of the whole ensemble: an
    uneven split (1 001 = 501 + 500), global trajectory numbering, fixed and adaptive (record NSAVED-1 per trajectory)."""
    from odefilters_jl_amd import host

    vf = orc.vector_field("lorenz63")
    N = 1001
    with host.DeviceGroup("lorenz63", 3, 1, N, 2, device_ids=[0, 0]) as grp:
        assert grp.shard(0) == (0, 501) and grp.shard(1) == (501, 500)
        grp.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
        if adaptive:
            grp.solve_adaptive(0.25, dt0=2.0**-9, max_steps=256)
        else:
            grp.solve_fixed(np.arange(33) * 2.0**-9)
        fin0 = grp.allgather(from_device=0)
        fin1 = grp.allgather(from_device=1)
    np.testing.assert_array_equal(fin0, fin1)
    ctx = pkg.Context("lorenz63", 3, 1, N)
    ctx.set_problem_perturbed(vf.u0, vf.p, 0.0, 1e-2)
    if adaptive:
        ctx.solve_adaptive(0.25, dt0=2.0**-9, max_steps=256)
        mean, ns = ctx.get(0), ctx.get(9)
        ref = np.stack([mean[ns[i] - 1, :, i] for i in range(N)], axis=1)
    else:
        ctx.solve_fixed(np.arange(33) * 2.0**-9)
        ref = ctx.get(0)[-1]
    np.testing.assert_array_equal(fin0, ref)
    ctx.close()
    # explicit u0 through the group: every shard takes its block
    u0s = orc.ensemble_u0(vf.u0, 9, 1e-2)
    with host.DeviceGroup("lorenz63", 3, 1, 9, 2, device_ids=[0, 0]) as grp:
        grp.set_problem(u0s, vf.p, 0.0)
        grp.solve_fixed(np.arange(9) * 2.0**-9)
        fin = grp.allgather()
    for i in (0, 4, 5, 8):
        r = orc.solve(vf, orc.EK1(order=3, smooth=False), u0=u0s[i], tspan=(0.0, 8 * 2.0**-9), dt=2.0**-9)
        np.testing.assert_allclose(fin[:3, i], r.u[-1], rtol=1e-11)


# ---- the Cholesky-failure branch (src/filtering.jl:38-47) -----------------------------------------------------------


@pytest.mark.parametrize("kernel", ["lane", "rows"])
def test_zero_predicted_covariance_constant_solution(pkg, kernel, monkeypatch):
    """u' = 0: z = 0, sigma^2 = 0, the predicted covariance is the zero matrix -- every Cholesky pivot and every
    reflector norm of the step vanishes.  The kernels' zero-pivot rule returns the exact constant solution with zero
    covariance (the reference's QR fallback gives the same zero factor and then fails in inv(S))."""
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", FILTER_KERNELS[kernel])
    monkeypatch.setenv("ODEF_SMOOTH_ROWS_MAX_N", FILTER_KERNELS[kernel])
    u0s = np.array([[0.75, -1.25], [2.0, 3.0], [1e-3, 1e3]])
    prob = pkg.EnsembleProblem(pkg.ODEProblem("linear", u0s[0], (0.0, 0.25), [0.0, 0.0]), u0s=u0s)
    for q in (1, 3, 5):
        sol = pkg.solve(prob, pkg.EK1(order=q), pkg.EnsembleHIP(), dt=2.0**-6, adaptive=False)
        assert sol.retcode == ["Success"] * 3
        m = sol.x_filt_mean()
        np.testing.assert_array_equal(m[:, :, :2], np.broadcast_to(u0s[:, None, :], m[:, :, :2].shape))
        assert np.all(m[:, :, 2:] == 0.0) and np.all(sol.x_filt_cov() == 0.0) and np.all(sol.x_smooth_cov() == 0.0)
        np.testing.assert_array_equal(sol.x_smooth_mean(), m)


def test_rank_deficient_predict_matches_the_qr_fallback(pkg):
    """odef_predict with a rank-deficient stacked factor [A L, Q_L] (L of rank 2 in dimension 6, Q_L = 0): the reference's
    cholesky! fails and `qr` supplies the factor (src/filtering.jl:42-46); the library zeroes the non-positive pivot's
    column.  Both must give the same predicted covariance A L L' A' (exactly singular)."""
    rng = np.random.default_rng(5)
    D, n = 6, 5
    A, _ = pkg.ibm(2, 2)
    L = np.zeros((n, D, D))
    L[:, :, :2] = rng.standard_normal((n, D, 2))  # rank 2
    mu = rng.standard_normal((n, D))
    mo, co = pkg.predict(mu, L, A, np.zeros((D, D)))
    for i in range(n):
        x = orc.predict(orc.SRGaussian(mu[i], L[i]), A, np.zeros((D, D)))
        np.testing.assert_allclose(mo[i], x.mu, rtol=1e-13, atol=1e-13)
        ref = x.L @ x.L.T
        np.testing.assert_allclose(co[i], ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        assert np.linalg.matrix_rank(co[i], tol=1e-9 * np.abs(ref).max()) == 2
