"""CPU tests of the oracle (oracle/sde_mpc_oracle.c): pins it against the committed golden vectors,
checks the adjoint against finite differences (float64 build) and basic properties."""
import ctypes as C

import numpy as np
import pytest

import orc
from cases import bits_differ, golden_cases, load_golden
from sde4mbrl_px4_amd import MPCConfig, synthetic_hexa, synthetic_iris
from sde4mbrl_px4_amd import workload as W


def test_spec_functions_accuracy():
    L = orc.lib()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(0)
    xs = (rng.standard_normal((20000, 4)) * 4).astype(np.float32)
    xs[:500] *= 5
    out = np.zeros_like(xs)
    for i in range(len(xs)):
        L.orc_tanh4(xs[i].ctypes.data_as(fp), out[i].ctypes.data_as(fp))
    assert np.abs(out - np.tanh(xs.astype(np.float64))).max() < 1e-6
    assert np.all(np.abs(out) <= 1.0 + 1e-6)   # approximation may overshoot 1 by < 1e-6
    x = np.linspace(-40, 40, 4001).astype(np.float32)
    s = np.array([L.orc_sigmoid(float(v)) for v in x])
    assert np.abs(s - 1 / (1 + np.exp(-x.astype(np.float64)))).max() < 2e-7
    d = np.exp(np.linspace(0, 70, 4001)).astype(np.float32)
    r = np.array([L.orc_rcp(float(v)) for v in d])
    assert np.abs(r * d.astype(np.float64) - 1).max() < 2e-7
    a = np.exp(np.linspace(-8, 8, 4001)).astype(np.float32)
    r = np.array([L.orc_rsqrt(float(v)) for v in a])
    assert np.abs(r * np.sqrt(a.astype(np.float64)) - 1).max() < 3e-7


@pytest.mark.parametrize("name", list(golden_cases().keys()))
def test_oracle_matches_golden(name):
    """The restatement reproduces its committed vectors bit for bit (guards against drift)."""
    cfg, model, seed, curr_t, pos = golden_cases()[name]
    g = load_golden(name)
    O = orc.Oracle(cfg, model)
    cost, traj, xmean = O.rollout(g["x0"], g["u"], g["xref"], g["noise"], True, True)
    assert np.float32(cost) == g["cost"]
    assert bits_differ(traj[:, -1, :], g["traj_last"]) == 0 and bits_differ(traj[0], g["traj_p0"]) == 0
    assert bits_differ(xmean, g["xmean"]) == 0
    gc, grad = O.grad(g["x0"], g["u"], g["xref"], g["noise"])
    assert np.float32(gc) == g["grad_cost"] and bits_differ(grad.astype(np.float32), g["grad"]) == 0
    xn, eta = O.step(g["x0"], g["u"][0], g["noise"][0, 0], 0)
    assert bits_differ(xn, g["step_xn"]) == 0 and np.float32(eta) == g["step_eta"]
    uopt, xevol, info, tr = O.solve(g["x0"], g["xref"], g["noise"], g["u_init"], float(g["stepsize_in"]), trace_cap=cfg.max_iter)
    assert bits_differ(uopt, g["uopt"]) == 0 and bits_differ(xevol, g["xevol"]) == 0 and bits_differ(info, g["info"]) == 0
    assert bits_differ(tr, g["trace"]) == 0
    # A8 post-processing (sde_control.py:428-432)
    thrust = uopt.sum(axis=1) / uopt.shape[1]
    wopt = np.stack([thrust, xevol[1:, 10], xevol[1:, 11], xevol[1:, 12]]).T
    np.testing.assert_array_equal(wopt.astype(np.float64), g["wopt"])


def _small_problem(H=8, P=5, m=4, seed=1, **kw):
    cfg = MPCConfig(horizon=H, num_short_dt=max(1, H // 2), long_step_dt=0.08, num_particles=P, u_slew_coeff=0.7, discount=0.95,
                    u_slew_constr=[[-0.05, 0.04]] * m, u_slew_constr_coeff=3.0, **kw)
    model = synthetic_iris(seed)
    x0 = W.random_initial_states(1, seed)[0]
    xref = W.reference_window(0.3, cfg.time_steps)
    noise = W.make_noise(1, P, H, seed)[0]
    rng = np.random.default_rng(seed)
    u = np.clip(0.71 + 0.1 * rng.standard_normal((H, m)), 1e-4, 1).astype(np.float32)
    return cfg, model, x0, xref, noise, u


def test_adjoint_matches_finite_differences_float64():
    cfg, model, x0, xref, noise, u = _small_problem()
    O64 = orc.Oracle(cfg, model, double=True)
    c, g = O64.grad(x0, u, xref, noise)
    u64 = u.astype(np.float64)
    fd = np.zeros_like(g)
    eps = 1e-6
    for t in range(cfg.horizon):
        for j in range(4):
            up, um = u64.copy(), u64.copy()
            up[t, j] += eps
            um[t, j] -= eps
            fd[t, j] = (O64.cost_du(x0, up, xref, noise) - O64.cost_du(x0, um, xref, noise)) / (2 * eps)
    assert np.abs(fd - g).max() <= 1e-6 * max(1.0, np.abs(g).max())


def test_float32_gradient_close_to_float64():
    cfg, model, x0, xref, noise, u = _small_problem(H=20, P=32)
    c32, g32 = orc.Oracle(cfg, model).grad(x0, u, xref, noise)
    c64, g64 = orc.Oracle(cfg, model, double=True).grad(x0, u, xref, noise)
    assert abs(c32 - c64) <= 1e-5 * abs(c64)
    assert np.abs(g32 - g64).max() <= 2e-4 * np.abs(g64).max()


def test_rollout_properties():
    cfg, model, x0, xref, noise, u = _small_problem(H=25, P=33)
    O = orc.Oracle(cfg, model)
    cost, traj, xmean = O.rollout(x0, u, xref, noise, True, True)
    assert np.isfinite(cost) and cost > 0
    qn = np.linalg.norm(traj[:, :, 6:10], axis=-1)
    assert np.abs(qn - 1).max() < 1e-5                       # quaternion renormalised every step
    np.testing.assert_array_equal(traj[:, 0, :], np.tile(x0, (33, 1)))
    np.testing.assert_allclose(xmean, traj.mean(axis=0), rtol=2e-5, atol=2e-6)
    # zero diffusion amplitude -> all particles identical
    model0 = synthetic_iris(1)
    model0.sigma = np.zeros(6, np.float32)
    _, traj0, _ = orc.Oracle(cfg, model0).rollout(x0, u, xref, noise, True, False)
    assert np.abs(traj0 - traj0[:1]).max() == 0.0
    # cost of identical particles does not depend on P (mean), up to rounding
    cfgp1 = cfg.replace(num_particles=1)
    c1, _, _ = orc.Oracle(cfgp1, model0).rollout(x0, u, xref, noise[:1], False, False)
    c33, _, _ = orc.Oracle(cfg, model0).rollout(x0, u, xref, noise, False, False)
    assert abs(c1 - c33) <= 1e-5 * abs(c1)


def test_solve_is_monotone_bounded_and_deterministic():
    cfg, model, x0, xref, noise, u = _small_problem(H=15, P=16, max_iter=30, max_no_improvement_iter=30)
    O = orc.Oracle(cfg, model)
    u0 = np.tile(np.float32(0.71), (15, 4))
    uopt, xevol, info, tr = O.solve(x0, xref, noise, u0, cfg.ls_init_stepsize, trace_cap=30)
    uopt2, xevol2, info2, _ = O.solve(x0, xref, noise, u0, cfg.ls_init_stepsize)
    assert bits_differ(uopt, uopt2) == 0 and bits_differ(info, info2) == 0
    assert uopt.min() >= 1e-4 and uopt.max() <= 1.0
    assert info[6] <= info[5]                                  # opt_cost <= init_cost
    c_opt, _, xm = O.rollout(x0, uopt, xref, noise, False, True)
    assert np.float32(c_opt) == info[6]                        # reported cost is the cost of the returned controls
    assert bits_differ(xm, xevol) == 0
    n_it = int(info[2])
    assert 1 <= n_it <= 30 and info[7] == tr[:n_it, 3].sum()
    assert abs(info[0] - tr[:n_it, 3].mean()) < 1e-5


def test_zero_iterations_returns_projected_warm_start():
    cfg, model, x0, xref, noise, u = _small_problem(H=6, P=4, max_iter=0)
    O = orc.Oracle(cfg, model)
    u0 = (u + 0.6).astype(np.float32)                          # partly outside [1e-4, 1]
    uopt, xevol, info, _ = O.solve(x0, xref, noise, u0, 0.01)
    np.testing.assert_array_equal(uopt, np.clip(u0, np.float32(1e-4), np.float32(1.0)))
    assert info[2] == 0 and info[5] == info[6]


def test_non_finite_gradient_terminates_without_a_step():
    """SPEC.md §8 guard: +inf cost / NaN gradient -> the solve returns the projected warm start, num_steps = 0, grad_sqr non-finite;
    a NaN in the warm start is projected to the lower bound (SPEC.md §3.6)."""
    from cases import diverging_single_rotor_case
    cfg, model, x0, xref, noise, u = diverging_single_rotor_case()
    O = orc.Oracle(cfg, model)
    c, g = O.grad(x0[1], u[1], xref[1], noise[1])
    assert np.isinf(c) and np.isnan(g).sum() >= 50
    uopt, xevol, info, _ = O.solve(x0[1], xref[1], noise[1], u[1], 0.01)
    np.testing.assert_array_equal(uopt, u[1])
    assert info[2] == 0 and not np.isfinite(info[3]) and np.isinf(info[5]) and np.isinf(info[6]) and info[7] == 0
    # the well-behaved instance next to it still iterates
    _, _, info2, _ = O.solve(x0[2], xref[2], noise[2], u[2], 0.01)
    assert info2[2] >= 1 and np.isfinite(info2[3])
    un = u[2].copy(); un[3, 0] = np.nan
    cfg0 = cfg.replace(max_iter=0)
    uo, _, _, _ = orc.Oracle(cfg0, model).solve(x0[2], xref[2], noise[2], un, 0.01)
    assert uo[3, 0] == np.float32(1e-4) and np.array_equal(np.delete(uo, 3, 0), np.delete(u[2], 3, 0))


def test_f16_rtz_quantiser_matches_numpy_semantics():
    L = orc.lib()
    L.orc_f16_rtz_value.argtypes, L.orc_f16_rtz_value.restype = [C.c_double], C.c_double
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.standard_normal(4000) * np.exp2(rng.integers(-28, 18, 4000)), [0.0, -0.0, 65504.0, 65520.0, 1e6, -1e6, 6.1e-5, 5.9e-8, 2.0 ** -25]]).astype(np.float32)
    for x in xs:
        q = np.float32(L.orc_f16_rtz_value(float(x)))
        with np.errstate(over="ignore"):
            h = np.float16(x)                               # round to nearest
        if np.isinf(h):
            h = np.float16(np.sign(x) * 65504.0)
        elif abs(np.float32(h)) > abs(x):                   # nearest went away from zero: step back toward zero
            h = np.nextafter(h, np.float16(0.0))
        assert q == np.float32(h), (x, q, h)
        assert abs(q) <= abs(x) and np.float32(np.float16(q)) == q


def test_f16_mode_is_close_to_f32_mode():
    cfg, model, x0, xref, noise, u = _small_problem(H=20, P=32)
    c32, g32 = orc.Oracle(cfg, model).grad(x0, u, xref, noise)
    c16, g16 = orc.Oracle(cfg.replace(mlp_dtype="f16"), model).grad(x0, u, xref, noise)
    assert c16 != c32                                       # the quantisation is really applied
    assert abs(c16 - c32) <= 1e-3 * abs(c32)
    assert np.abs(g16 - g32).max() <= 2e-2 * np.abs(g32).max()


def test_worker_replay_fixture_is_reproduced_by_the_oracle_backed_worker():
    """tests/golden/worker_replay.npz (SURVEY.md §8c last row, §8f N1): the committed fixture is what the worker loop (worker.py =
    sde_control.py:365-450) produces today when its solver is the CPU oracle — guards the fixture against drift of either."""
    import replay
    fx = dict(np.load(replay.FIXTURE))
    got = replay.run(oracle=True)
    assert set(got) == set(fx)
    for k in fx:
        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(fx[k])
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k
    assert len(replay.MODES) >= 30 and fx["sel_idx"].max() == 11 and fx["sel_idx"].min() == 0


def test_vectorised_timing_build_agrees_with_the_checker_build():
    """oracle/liborc_vec.so (-DORC_VEC: the same source, 16 particles per call, contraction allowed) is bench.py's CPU timing leg, never the
    checker: a full solve must agree with the bit-exact scalar build to well inside the north star's 1e-4 on the controls, ragged particle
    counts included (dead lanes of the last block)."""
    from sde4mbrl_px4_amd import MPCConfig
    for P, m, seed in ((40, 4, 1), (17, 6, 2), (1, 4, 3), (128, 4, 4)):
        kw = dict(horizon=10, num_short_dt=6, long_step_dt=0.1, num_particles=P, u_slew_coeff=1.0, max_iter=8, max_no_improvement_iter=8)
        if m == 6:
            kw.update(input_id=list(range(6)), input_bound=[[1e-4, 1.0]] * 6, uref=[0.42] * 6)
        cfg = MPCConfig(**kw)
        model = synthetic_iris() if m == 4 else synthetic_hexa()
        x0 = W.random_initial_states(1, seed)[0]
        xref = W.reference_window(0.2, cfg.time_steps)
        noise = W.make_noise(1, P, 10, seed)[0]
        u = np.tile(np.asarray(cfg.uref, np.float32), (10, 1))
        a = orc.Oracle(cfg, model).solve(x0, xref, noise, u, 0.01)
        b = orc.Oracle(cfg, model, vec=True).solve(x0, xref, noise, u, 0.01)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b[1], a[1], rtol=1e-4, atol=1e-4)
        assert b[2][2] == a[2][2] and b[2][7] == a[2][7]                  # same iteration and line-search counts
        np.testing.assert_allclose(b[2][5:7], a[2][5:7], rtol=1e-5)


def test_state_constr_penalty_is_active_and_differentiated():
    """SPEC.md §5.3 state bounds (iris_sitl_traj_mpc.yaml:16-29, penalty form): the term raises the cost when the rollout leaves the box, leaves it
    unchanged when the box is wide, and its gradient matches central differences of the float64 build."""
    kw = dict(horizon=8, num_short_dt=8, num_particles=12, u_slew_coeff=1.0)
    sc = dict(state_id=[3, 4, 5, 10, 11, 12], state_penalty=[10.0, 10.0, 20.0, 10.0, 10.0, 10.0], constr_pen=0.1)
    tight = MPCConfig(**kw, **sc, state_bound=[[-0.05, 0.05]] * 6)
    wide = MPCConfig(**kw, **sc, state_bound=[[-1e6, 1e6]] * 6)
    none = MPCConfig(**kw)
    model = synthetic_iris()
    x0 = W.random_initial_states(1, 3)[0]
    xref = W.reference_window(0.1, none.time_steps)
    noise = W.make_noise(1, 12, 8, 9)[0]
    u = np.clip(0.71 + 0.1 * np.random.default_rng(1).standard_normal((8, 4)), 1e-4, 1).astype(np.float32)
    c_none = orc.Oracle(none, model).rollout(x0, u, xref, noise)[0]
    c_wide = orc.Oracle(wide, model).rollout(x0, u, xref, noise)[0]
    c_tight = orc.Oracle(tight, model).rollout(x0, u, xref, noise)[0]
    assert c_wide == c_none and c_tight > c_none + 0.1                    # (the tracking terms dominate this rollout: 436.3 -> 436.7)
    Od = orc.Oracle(tight, model, double=True)
    _, g = Od.grad(x0, u, xref, noise)
    u64 = u.astype(np.float64)
    for (t, j) in ((0, 0), (3, 2), (7, 3)):
        up, um = u64.copy(), u64.copy()
        up[t, j] += 1e-6; um[t, j] -= 1e-6
        fd = (Od.cost_du(x0, up, xref, noise) - Od.cost_du(x0, um, xref, noise)) / 2e-6
        assert abs(fd - g[t, j]) <= 1e-5 * max(1.0, abs(g[t, j])), (t, j, fd, g[t, j])
