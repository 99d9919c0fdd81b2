"""Host mirror of the reference's solver interface: argument handling that needs no GPU."""
import numpy as np
import pytest


def test_scalar_u0_rejected(pkg):
    """test/errors.jl:11-14 / src/caches.jl:46-49."""
    with pytest.raises(pkg.OdefError, match="scalar- or vector-valued"):
        pkg.ODEProblem("linear", 1.0, (0.0, 1.0), (1.0, 1.0))
    with pytest.raises(pkg.OdefError):
        pkg.ODEProblem("linear", np.ones((2, 2)), (0.0, 1.0), (1.0, 1.0))


def test_fixed_steps_need_dt(pkg):
    """test/errors.jl:17-19."""
    prob = pkg.ODEProblem("fhn", [-1.0, 1.0], (0.0, 20.0), (0.2, 0.2, 3.0))
    with pytest.raises(pkg.OdefError, match="Fixed timestep methods require a choice of dt"):
        pkg.solve(prob, pkg.EK0(order=1), adaptive=False)


def test_alg_defaults_match_reference(pkg):
    """src/algorithms.jl:23-28,46-51."""
    for Alg in (pkg.EK0, pkg.EK1):
        a = Alg()
        assert (a.prior, a.order, a.diffusionmodel, a.smooth) == ("ibm", 3, "dynamic", True)
    with pytest.raises(pkg.OdefError, match="ibm prior"):
        pkg.solve(pkg.ODEProblem("fhn", [-1.0, 1.0], (0.0, 1.0), (0.2, 0.2, 3.0)), pkg.EK0(prior="ioup"), dt=0.1, adaptive=False)


def test_fixed_time_grid(pkg, orc):
    g = pkg.fixed_time_grid(0.0, 20.0, 7e-2)
    assert len(g) == 287 and g[-1] == 20.0 and abs((g[-1] - g[-2]) - 0.05) < 1e-9
    np.testing.assert_array_equal(g, orc.fixed_time_grid(0.0, 20.0, 7e-2))
    g2 = pkg.fixed_time_grid(0.0, 2.0, 2.0**-9)
    np.testing.assert_array_equal(g2, np.arange(1025) * 2.0**-9)


def test_unpack_tril(pkg):
    D = 4
    M = np.arange(16.0).reshape(4, 4)
    M = M + M.T
    packed = np.array([M[i, j] for i in range(D) for j in range(i + 1)])
    np.testing.assert_array_equal(pkg.unpack_tril(packed[None], D)[0], M)


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


def test_user_vector_field_compiles_without_a_gpu(pkg):
    """odef_rhs_compile: a hipcc child process cross-compiles the lane kernels around a user vector field for gfx950 (no
    GPU needed to compile); a text that does not provide the interface is rejected with the compiler log, and so is a
    state dimension whose kernels the register allocator cannot place -- as an error, not as a crash."""
    name = pkg.compile_rhs("UserLorenzCPU", USER_LORENZ, 3, 3, struct_name="UserLorenz")
    import importlib

    h = importlib.import_module(pkg.Context.__module__)
    assert h.RHS[name] >= 100 and h.RHS_DIMS[name] == (3, 3)
    with pytest.raises(pkg.OdefError, match="no member named 'f'"):
        pkg.compile_rhs("Broken", "struct Broken { static constexpr int d = 2, np = 0; };", 2, 0)
    with pytest.raises(pkg.OdefError, match="d of the struct differs"):
        pkg.compile_rhs("WrongDim", USER_LORENZ.replace("UserLorenz", "WrongDim"), 2, 3)
    with pytest.raises(pkg.OdefError, match="d must be in 1..32"):
        pkg.compile_rhs("TooBig", USER_LORENZ.replace("UserLorenz", "TooBig"), 40, 3)
    # above d = 10 no lane kernel exists to try the text on: a probe kernel (f in double, in forward mode, on Taylor jets) does
    l96 = "struct L96d12 { static constexpr int d = 12, np = 1; template <class T> __device__ static void f(const T (&u)[12], const double* p, T (&du)[12]) {" \
          " for (int i = 0; i < 12; ++i) du[i] = (u[(i + 1) % 12] - u[(i + 10) % 12]) * u[(i + 11) % 12] - u[i] + p[0]; } };"
    assert h.RHS[pkg.compile_rhs("L96d12", l96, 12, 1)] >= 100
    with pytest.raises(pkg.OdefError, match="d of the struct differs"):
        pkg.compile_rhs("L96d12b", l96.replace("L96d12", "L96d12b"), 14, 1)
    # a device function the compiler keeps out of line would be shared by kernels with different register budgets (a fault
    # on the GPU, nothing at compile time): the code object is inspected and refused
    helper = "__device__ __attribute__((noinline)) double odef_test_outlined(double x) { return 1.5 * x; }\n"
    text = helper + USER_LORENZ.replace("UserLorenz", "Outlined").replace("const double s = p[0],", "const double s = odef_test_outlined(p[0]) / 1.5,")
    assert text.count("odef_test_outlined(p[0])") == 2
    with pytest.raises(pkg.OdefError, match="out of line"):
        pkg.compile_rhs("Outlined", text, 3, 3)


def test_fixed_time_grid_validation_and_tstops(pkg):
    """`dt <= 0` or a decreasing time span must raise instead of looping forever; `tstops` are hit exactly
    and combine with `dt` as in OrdinaryDiffEq's loop."""
    from odefilters_jl_amd import host

    for bad in (0.0, -1e-3, float("nan"), float("inf")):
        with pytest.raises(host.OdefError):
            host.fixed_time_grid(0.0, 1.0, bad)
    with pytest.raises(host.OdefError):
        host.fixed_time_grid(1.0, 1.0, 0.1)
    g = host.fixed_time_grid(0.0, 1.0, 0.25)
    np.testing.assert_array_equal(g, [0.0, 0.25, 0.5, 0.75, 1.0])
    g = host.fixed_time_grid(0.0, 1.0, 0.25, tstops=[0.3, 0.5, 2.0])
    np.testing.assert_allclose(g, [0.0, 0.25, 0.3, 0.5, 0.75, 1.0], rtol=0, atol=1e-15)
    assert 0.3 in g and 0.5 in g and g[-1] == 1.0
    g = host.fixed_time_grid(0.0, 2.0, 2.0**-9)
    assert len(g) == 1025 and np.all(np.diff(g) == 2.0**-9)  # BASELINE grid: exact binary steps, no clipping
