"""BASELINE.json configurations 2-5 at their FULL sizes on the GPU: size-independent properties over the whole ensemble
plus, for a handful of trajectories, value-by-value comparison with oracle fixtures computed at the full step count
(tests/golden/full_*.npz, generator tests/golden/make_fullsize.py).  The kernels a configuration gets by default are the
ones under test (row-team kernels for configs 2 and 5, the lane kernel for config 3, the tiled kernel for config 4).

Tolerances: solution components u = E0 mu at 1e-10 relative (BASELINE target 1e-8); the derivative blocks and the
covariances are ill-conditioned in the reference's own arithmetic (tests/_parity.py) and are compared at the level the
float64 oracle itself reproduces under 1-ulp input changes (1e-6 of the block scale, 5e-3 of the covariance scale at the
transient near t = 1.15, tests/golden/exact_lorenz_mp.npz)."""
import os

import numpy as np
import pytest

import _parity as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LORENZ_U0, LORENZ_P = [1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0]


def _block_tol_check(got, ref, d, what):
    be = P.block_err(got, ref, d)
    assert be[0] <= 1e-10, f"{what}: u block off by {be[0]:.2e}"
    assert np.all(be[1:] <= 1e-6), f"{what}: derivative blocks off by {be}"


def test_config2_lorenz_4096_filter_and_smoother(pkg):
    """configs[1]: Lorenz-63 EK1(3), 4 096 trajectories x 1 024 steps, every step saved, RTS smoother."""
    fx = np.load(os.path.join(GOLD, "full_lorenz_fixed.npz"))
    N, ns, dt = 4096, int(fx["nsteps"]), float(fx["dt"])
    ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
    ctx.set_problem_perturbed(LORENZ_U0, LORENZ_P, 0.0, 1e-2)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    ctx.smooth()
    assert (ctx.get(10) == 0).all() and (ctx.get(9) == ns + 1).all()
    mean, smean = ctx.get(0), ctx.get(11)  # [n_save, D, N]
    cov, scov = ctx.get(1), ctx.get(12)
    assert np.isfinite(mean).all() and np.isfinite(smean).all() and np.isfinite(cov).all() and np.isfinite(scov).all()
    assert np.all(cov[0] == 0.0) and np.all(scov[0] == 0.0)  # x0 is exact (test/solution.jl:38-41)
    np.testing.assert_array_equal(smean[-1], mean[-1])  # the last smoothed state is the last filter state (test/smoothing.jl)
    np.testing.assert_array_equal(scov[-1], cov[-1])
    # PSD over the whole ensemble at three times (filter and smoothed), variances decrease under smoothing
    for s in (1, 512, ns):
        for c in (cov, scov):
            w = np.linalg.eigvalsh(pkg.unpack_tril(c[s].T, 12))
            assert w.min() > -1e-9 * np.abs(w).max()
    diag = [k * (k + 1) // 2 + k for k in range(12)]
    assert np.all(scov[1:ns][:, diag] <= cov[1:ns][:, diag] * (1 + 1e-9) + 1e-300)
    steps = fx["steps"]
    for k, gi in enumerate(fx["idx"]):
        if gi >= N:
            continue
        _block_tol_check(mean[steps][:, :, gi], fx["mean_filt"][k], 3, f"config 2 filter, trajectory {gi}")
        _block_tol_check(smean[steps][:, :, gi], fx["mean_smooth"][k], 3, f"config 2 smoother, trajectory {gi}")
        assert P.cov_err(pkg.unpack_tril(cov[steps][:, :, gi], 12), fx["cov_filt"][k]) < 5e-3
        assert P.cov_err(pkg.unpack_tril(scov[steps][:, :, gi], 12), fx["cov_smooth"][k]) < 5e-3
        # sigma^2 = z' W^-1 z / d is a squared RESIDUAL: it inherits the noise of the highest derivative twice over
        np.testing.assert_allclose(ctx.get(2)[1:, gi], fx["diffusions"][k], rtol=2e-3)
        np.testing.assert_allclose(ctx.get(4)[gi], fx["loglik"][k], rtol=1e-6)
    # trajectory 0 IS the trajectory of the 50-digit evaluation (tests/golden/exact_lorenz_mp.npz, all 1 025 records, filter and
    # smoother): there the bar is not a flat number but the float64 oracle's own distance from the exact result, block by block
    ex = np.load(os.path.join(GOLD, "exact_lorenz_mp.npz"))
    np.testing.assert_array_equal(ctx.get(13)[:, 0], ex["u0"])
    P.check_against_exact(mean[:, :, 0], pkg.unpack_tril(cov[:, :, 0], 12), ex["mean_filt"], ex["cov_filt"], ex["oracle_block_err_filt"],
                          ex["oracle_cov_err_filt"], 3, "config 2 filter, trajectory 0 against the 50-digit evaluation")
    P.check_against_exact(smean[:, :, 0], pkg.unpack_tril(scov[:, :, 0], 12), ex["mean_smooth"], ex["cov_smooth"], ex["oracle_block_err_smooth"],
                          ex["oracle_cov_err_smooth"], 3, "config 2 smoother, trajectory 0 against the 50-digit evaluation")
    # final-only save mode = the last every-step record, bit for bit
    ctx2 = pkg.Context("lorenz63", 3, 1, N, save_everystep=False)
    ctx2.set_problem_perturbed(LORENZ_U0, LORENZ_P, 0.0, 1e-2)
    ctx2.solve_fixed(np.arange(ns + 1) * dt)
    np.testing.assert_array_equal(ctx2.get(0)[0], mean[-1])
    np.testing.assert_array_equal(ctx2.get(1)[0], cov[-1])
    ctx.close()
    ctx2.close()


def test_config3_lorenz_65536_final_state(pkg):
    """configs[2]: 65 536 trajectories x 1 024 steps (final-save mode here: the every-step record of this size is
    48.9 GB and is what bench.py times and checks); duplicated inputs give bitwise equal outputs."""
    fx = np.load(os.path.join(GOLD, "full_lorenz_fixed.npz"))
    N, ns, dt = 65536, int(fx["nsteps"]), float(fx["dt"])
    ctx = pkg.Context("lorenz63", 3, 1, N, save_everystep=False)
    ctx.set_problem_perturbed(LORENZ_U0, LORENZ_P, 0.0, 1e-2)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    assert (ctx.get(10) == 0).all()
    mean, cov = ctx.get(0)[0], ctx.get(1)[0]
    assert np.isfinite(mean).all() and np.isfinite(cov).all()
    for k, gi in enumerate(fx["idx"]):
        _block_tol_check(mean[None, :, gi], fx["mean_filt"][k][-1:], 3, f"config 3, trajectory {gi}")
        assert P.cov_err(pkg.unpack_tril(cov[None, :, gi], 12), fx["cov_filt"][k][-1:]) < 5e-3
    w = np.linalg.eigvalsh(pkg.unpack_tril(cov[:, ::64].T, 12))
    assert w.min() > -1e-9 * np.abs(w).max()
    # the 8-GPU shard of the same ensemble (8 192 trajectories, row-team kernel): same trajectories to rounding
    ctx8 = pkg.Context("lorenz63", 3, 1, 8192, save_everystep=False)
    ctx8.set_problem_perturbed(LORENZ_U0, LORENZ_P, 0.0, 1e-2)
    ctx8.solve_fixed(np.arange(ns + 1) * dt)
    np.testing.assert_allclose(ctx8.get(0)[0][:3], mean[:3, :8192], rtol=1e-10)
    ctx.close()
    ctx8.close()


def test_config4_pleiades_8192_final_state(pkg):
    """configs[3]: Pleiades d = 28, EK1(5) (state dimension 168), 8 192 trajectories x 256 steps, final state."""
    fx = np.load(os.path.join(GOLD, "full_pleiades.npz"))
    N, ns, dt = 8192, int(fx["nsteps"]), float(fx["dt"])
    base = np.array([3, 3, -1, -3, 2, -2, 2, 3, -3, 2, 0, 0, -4, 4, 0, 0, 0, 0, 0, 1.75, -1.5, 0, 0, 0, -1.25, 1, 0, 0], float)
    ctx = pkg.Context("pleiades", 5, 1, N, save_everystep=False)
    ctx.set_problem_perturbed(base, [], 0.0, 1e-3, n_perturbed=14)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    assert (ctx.get(10) == 0).all()
    mean, cov = ctx.get(0)[0], ctx.get(1)[0]
    assert np.isfinite(mean).all() and np.isfinite(cov).all()
    diag = np.array([k * (k + 1) // 2 + k for k in range(168)])
    assert (cov[diag] >= 0).all()
    # positive semi-definite after 256 Joseph-form steps on the matrix cores (T - E K' is not PSD by construction):
    # the fixture's trajectories and a spread of others, eigenvalues on the scale of the largest one
    for gi in sorted(set(int(g) for g in fx["idx"]) | set(range(0, N, 547))):
        w = np.linalg.eigvalsh(pkg.unpack_tril(cov[:, gi][None], 168)[0])
        assert w.min() >= -1e-9 * np.abs(w).max(), (gi, w.min(), w.max())
    for k, gi in enumerate(fx["idx"]):
        np.testing.assert_allclose(ctx.get(13)[:, gi], fx["u0s"][k], rtol=0, atol=0)  # same ensemble as the fixture's
        np.testing.assert_allclose(mean[:28, gi], fx["u_final"][k], rtol=1e-10)
        be = P.block_err(mean[None, :, gi], fx["mean_final"][k][None], 28)
        assert be[0] <= 1e-10 and np.all(be[1:4] <= 1e-6), be
        # variances: the position / velocity blocks sit at eps^2 (1e-32, pure rounding), so they are compared on the
        # scale of the largest variance of the state
        # ... and the two highest derivative blocks are the ill-conditioned ones (oracle vs extended precision after 24
        # steps: 1e-3 in the covariance, tests/golden/exact_pleiades_ld.npz; 256 steps here)
        vref = fx["var_final"][k]
        np.testing.assert_allclose(cov[diag[:112], gi], vref[:112], rtol=2e-3, atol=1e-6 * vref.max())
        np.testing.assert_allclose(cov[diag[112:], gi], vref[112:], rtol=5e-2)
    ctx.close()


def test_pleiades_1024_every_step_and_smoother_properties(pkg):
    """The D = 168 every-step filter and smoother at a size where the record stage (csrc/record_stage.h) works on gigabytes
    (1 024 trajectories x 32 steps x 113 KB per record, filter records and smoothed records through it): properties that
    do not need the oracle -- every record finite with non-negative variances, the last smoothed record IS the last filter
    record, duplicated inputs give bit-identical outputs wherever they sit in the ensemble, the every-step solve ends in
    the final-state solve's record, and smoothing does not increase the variances of the solution block."""
    N, ns, dt = 1024, 32, 2.0**-10
    base = np.array([3, 3, -1, -3, 2, -2, 2, 3, -3, 2, 0, 0, -4, 4, 0, 0, 0, 0, 0, 1.75, -1.5, 0, 0, 0, -1.25, 1, 0, 0], float)
    rng = np.random.default_rng(5)
    u0 = base[:, None] + 1e-3 * np.concatenate([rng.standard_normal((14, N)), np.zeros((14, N))])
    u0[:, 777] = u0[:, 3]  # duplicates far apart (different workgroups, different XCDs)
    u0[:, 1023] = u0[:, 64]
    ctx = pkg.Context("pleiades", 3, 1, N, smooth=True)
    ctx.set_problem(u0.T.copy(), [], 0.0)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    ctx.smooth()
    assert (ctx.get(10) == 0).all()
    mean, cov, smean, scov = ctx.get(0), ctx.get(1), ctx.get(11), ctx.get(12)
    D = 112
    diag = np.array([k * (k + 1) // 2 + k for k in range(D)])
    for a in (mean, cov, smean, scov):
        assert np.isfinite(a).all()
    assert (cov[:, diag] >= 0).all() and (scov[:, diag] >= 0).all()
    np.testing.assert_array_equal(smean[-1], mean[-1])
    np.testing.assert_array_equal(scov[-1], cov[-1])
    np.testing.assert_array_equal(scov[0], cov[0])
    for a in (mean, cov, smean, scov):
        np.testing.assert_array_equal(a[:, :, 777], a[:, :, 3])
        np.testing.assert_array_equal(a[:, :, 1023], a[:, :, 64])
    # smoothing conditions on more data: the variances of u do not grow (up to rounding on the scale of the record)
    vu_f, vu_s = cov[1:-1, diag[:28]], scov[1:-1, diag[:28]]
    assert (vu_s <= vu_f * (1 + 1e-9) + 1e-30).all()
    ctx.close()
    ctx2 = pkg.Context("pleiades", 3, 1, N, save_everystep=False)
    ctx2.set_problem(u0.T.copy(), [], 0.0)
    ctx2.solve_fixed(np.arange(ns + 1) * dt)
    np.testing.assert_array_equal(ctx2.get(0)[0], mean[-1])
    np.testing.assert_array_equal(ctx2.get(1)[0], cov[-1])
    ctx2.close()


def test_pleiades_order5_smoother_properties_on_chip_pass(pkg):
    """The BASELINE order of Pleiades (q = 5, D = 168: eleven tile rows, the single-buffered product scheme and the half-filled
    last tile row of csrc/smooth_onchip.h) through the every-step filter and the split-pass smoother at 512 trajectories x 16 steps:
    properties that need no oracle -- finite records, the last smoothed record is the last filter record, duplicated inputs give
    bit-identical outputs wherever they sit, smoothed covariances positive semi-definite and not larger than the filter's
    (Sigma^f - Sigma^s = G (Sigma^- - Sigma^s_+) G' is positive semi-definite in exact arithmetic) on a spread of trajectories
    and records."""
    N, ns, dt = 512, 16, 2.0**-7
    base = np.array([3, 3, -1, -3, 2, -2, 2, 3, -3, 2, 0, 0, -4, 4, 0, 0, 0, 0, 0, 1.75, -1.5, 0, 0, 0, -1.25, 1, 0, 0], float)
    rng = np.random.default_rng(11)
    u0 = base[:, None] + 1e-3 * np.concatenate([rng.standard_normal((14, N)), np.zeros((14, N))])
    u0[:, 300] = u0[:, 5]
    u0[:, 511] = u0[:, 64]
    ctx = pkg.Context("pleiades", 5, 1, N, smooth=True)
    ctx.set_problem(u0.T.copy(), [], 0.0)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    ctx.smooth()
    assert (ctx.get(10) == 0).all()
    assert "rts_smooth_sweeps_kernel<28, 5>" in ctx.kernel_name(1)
    mean, cov, smean, scov = ctx.get(0), ctx.get(1), ctx.get(11), ctx.get(12)
    for a in (mean, cov, smean, scov):
        assert np.isfinite(a).all()
    np.testing.assert_array_equal(smean[-1], mean[-1])
    np.testing.assert_array_equal(scov[-1], cov[-1])
    np.testing.assert_array_equal(scov[0], cov[0])
    for a in (mean, cov, smean, scov):
        np.testing.assert_array_equal(a[:, :, 300], a[:, :, 5])
        np.testing.assert_array_equal(a[:, :, 511], a[:, :, 64])
    for i in (0, 77, 256, 511):
        for s_ in (1, ns // 2, ns - 1):
            cs, cf = pkg.unpack_tril(scov[s_][:, i], 168), pkg.unpack_tril(cov[s_][:, i], 168)
            # compare in preconditioned coordinates (the blocks of the state differ by h^-1 per derivative: src/preconditioning.jl)
            p = np.repeat(dt ** (np.arange(6) - 5.5), 28)
            cs, cf = cs * np.outer(p, p), cf * np.outer(p, p)
            scale = np.linalg.eigvalsh(cf).max()
            assert np.linalg.eigvalsh(cs).min() >= -1e-9 * scale, (i, s_)
            assert np.linalg.eigvalsh(cf - cs).min() >= -1e-9 * scale, (i, s_)
    ctx.close()


def test_config5_lorenz_16384_adaptive_and_smoother(pkg):
    """configs[4]: Lorenz-63 EK1(3), 16 384 trajectories, adaptive PI step-size control + RTS smoothing."""
    fx = np.load(os.path.join(GOLD, "full_lorenz_adaptive.npz"))
    N = 16384
    ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
    ctx.set_problem_perturbed(LORENZ_U0, LORENZ_P, 0.0, 1e-2)
    ctx.solve_adaptive(2.0, 1e-6, 1e-3, 2.0**-9, max_steps=400)
    ctx.smooth()
    ret, nsaved, nacc, nrej = ctx.get(10), ctx.get(9), ctx.get(5), ctx.get(6)
    assert (ret == 0).all() and (nsaved == nacc + nrej + 1).all() and nsaved.max() <= 401
    T, mean, smean, scov = ctx.get(3), ctx.get(0), ctx.get(11), ctx.get(12)
    last = nsaved - 1
    cols = np.arange(N)
    assert np.all(T[last, cols] == 2.0) and np.all(T[0] == 0.0)
    assert np.all(np.diff(T, axis=0)[: nsaved.min() - 1] >= 0)
    np.testing.assert_array_equal(smean[last, :, cols], mean[last, :, cols])  # last smoothed = last filter state
    assert np.isfinite(smean[0]).all() and np.isfinite(scov[1]).all()
    for k, gi in enumerate(fx["idx"]):
        n = nsaved[gi]
        t = T[:n, gi]
        keep = np.concatenate([[True], t[1:] != t[:-1]])  # drop the repeated records of rejected attempts
        tt, mf, msm = t[keep], mean[:n, :, gi][keep], smean[:n, :, gi][keep]
        assert (nacc[gi], nrej[gi]) == tuple(fx[f"counts{k}"]), f"trajectory {gi}: accepted/rejected differ from the oracle"
        np.testing.assert_allclose(tt, fx[f"t{k}"], rtol=1e-9)
        np.testing.assert_allclose(mf[:, :3], fx[f"mean_filt{k}"][:, :3], rtol=1e-7)
        np.testing.assert_allclose(msm[:, :3], fx[f"mean_smooth{k}"][:, :3], rtol=1e-7)
        var = scov[:n, :, gi][keep][:, [0, 2, 5]]
        np.testing.assert_allclose(var[1:], fx[f"var_smooth{k}"][1:, :3], rtol=1e-3)
    ctx.close()


# ---- against EXACT evaluations of the reference algorithm (tests/golden/make_exact.py) --------------------------------


@pytest.mark.parametrize("kernel", ["lane", "rows"])
def test_lorenz_1024_steps_against_50_digit_evaluation(pkg, kernel, monkeypatch):
    """Trajectory 0 of the BASELINE ensemble over all 1 024 steps, filter and smoother, both kernel families through the
    C ABI: |device - exact| <= 16 x |float64 oracle - exact| per derivative block and for the covariance, where exact is
    the 50-digit mpmath evaluation of the reference algorithm (tests/golden/exact_lorenz_mp.npz)."""
    v = {"lane": "0", "rows": "1000000000"}[kernel]
    monkeypatch.setenv("ODEF_FILTER_ROWS_MAX_N", v)
    monkeypatch.setenv("ODEF_SMOOTH_ROWS_MAX_N", v)
    if kernel == "lane":
        monkeypatch.setenv("ODEF_SMOOTH_LANE_MIN_N", "1")
    fx = np.load(os.path.join(GOLD, "exact_lorenz_mp.npz"))
    N, ns, dt = 16, int(fx["nsteps"]), float(fx["dt"])
    ctx = pkg.Context("lorenz63", 3, 1, N, smooth=True)
    ctx.set_problem(np.tile(fx["u0"], (N, 1)), LORENZ_P, 0.0)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    ctx.smooth()
    mean, cov, smean, scov = ctx.get(0), ctx.get(1), ctx.get(11), ctx.get(12)
    for i in (0, N - 1):
        P.check_against_exact(mean[:, :, i], pkg.unpack_tril(cov[:, :, i], 12), fx["mean_filt"], fx["cov_filt"],
                              fx["oracle_block_err_filt"], fx["oracle_cov_err_filt"], 3, f"{kernel} filter [{i}]")
        P.check_against_exact(smean[:, :, i], pkg.unpack_tril(scov[:, :, i], 12), fx["mean_smooth"], fx["cov_smooth"],
                              fx["oracle_block_err_smooth"], fx["oracle_cov_err_smooth"], 3, f"{kernel} smoother [{i}]")
    np.testing.assert_array_equal(mean[:, :, 0], mean[:, :, N - 1])  # same input, same bits, wherever it sits
    ctx.close()


def test_pleiades_24_steps_against_extended_precision(pkg):
    fx = np.load(os.path.join(GOLD, "exact_pleiades_ld.npz"))
    ns, dt = int(fx["nsteps"]), float(fx["dt"])
    ctx = pkg.Context("pleiades", 5, 1, 2)
    ctx.set_problem(np.tile(fx["u0"], (2, 1)), [], 0.0)
    ctx.solve_fixed(np.arange(ns + 1) * dt)
    mean, cov = ctx.get(0), ctx.get(1)
    be = P.block_err(mean[:, :, 1], fx["mean_filt"], 28)
    assert be[0] <= P.U_RTOL
    assert np.all(be <= np.maximum(P.EXACT_FACTOR * fx["oracle_block_err_filt"], 1e-15)), (be, fx["oracle_block_err_filt"])
    assert P.cov_err(pkg.unpack_tril(cov[-1:, :, 1], 168), fx["cov_final"][None]) <= P.EXACT_FACTOR * float(fx["oracle_cov_err_final"])
    ctx.close()
