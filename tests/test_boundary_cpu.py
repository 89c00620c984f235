"""CPU tests of the Python boundary (host logic only: no solver calls)."""
import os

import numpy as np
import pytest

from cases import CDIR
from sde4mbrl_px4_amd import jax_shim
from sde4mbrl_px4_amd import synthetic_iris
from sde4mbrl_px4_amd.sde_mpc_design import OptState, load_mpc_from_cfgfile
from sde4mbrl_px4_amd.utils import TrajectoryCSV, enu2ned
from worker import CONTROL_STATE, SharedBlocks, select_command
from sde4mbrl_px4_amd.workload import HOVER, lemniscate_state, random_initial_states


def test_enu2ned_is_an_involution_and_maps_axes():
    for x in random_initial_states(5, 3):
        y = enu2ned(x, np)
        assert y.dtype == np.float32 and y.shape == (13,)
        z = enu2ned(y, np)
        np.testing.assert_allclose(z[:6], x[:6], atol=1e-6)
        np.testing.assert_allclose(z[10:], x[10:], atol=1e-6)
        assert min(np.abs(z[6:10] - x[6:10]).max(), np.abs(z[6:10] + x[6:10]).max()) < 1e-6
        np.testing.assert_allclose(np.linalg.norm(y[6:10]), 1.0, atol=1e-6)
    e = HOVER.copy()
    e[0:3] = [1, 2, 3]
    np.testing.assert_allclose(enu2ned(e, np)[0:3], [2, 1, -3])
    # level, body-x pointing East in ENU  ==  level, heading +90 deg (East) in NED
    q = enu2ned(HOVER, np)[6:10]
    np.testing.assert_allclose(np.abs(q), [np.sqrt(0.5), 0, 0, np.sqrt(0.5)], atol=1e-6)
    assert q[0] * q[3] > 0


def test_trajectory_csv_interpolation(tmp_path):
    p = tmp_path / "traj.csv"
    p.write_text("t,x,y,z,vx,vy,vz,ax,ay,az,yaw,extra\n0,0,0,1,1,0,0,0,0,0,0,9\n1,1,0,1,1,0,0,0,0,0,1.0,9\n3,3,2,1,1,1,0,0,0,0,1.0,\n")
    tr = TrajectoryCSV(str(p))
    s = tr.state(0.5)
    np.testing.assert_allclose(s[:6], [0.5, 0, 1, 1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(s[6:10], [np.cos(0.25), 0, 0, np.sin(0.25)], atol=1e-6)
    np.testing.assert_allclose(tr.state(2.0)[:6], [2, 1, 1, 1, 0.5, 0], atol=1e-6)
    np.testing.assert_allclose(tr.state(-1.0)[:3], [0, 0, 1])          # clamped before the first sample
    np.testing.assert_allclose(tr.state(10.0)[:3], [3, 2, 1])          # last setpoint after the end (geometric_controller.cpp:224-236)
    assert tr.state(np.array([0.0, 0.5, 1.0])).shape == (3, 13)
    (tmp_path / "bad.csv").write_text("t,x,y\n0,0,0\n")
    with pytest.raises(ValueError, match="lacks columns"):
        TrajectoryCSV(str(tmp_path / "bad.csv"))


def test_load_mpc_from_cfgfile_shapes_without_touching_the_gpu():
    cfg_dict, (m_reset, m_mpc), sft, extra = load_mpc_from_cfgfile(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), convert_to_enu=True)
    assert extra is None and sft is None                                 # position controller: no trajectory (sde_control.py:177)
    assert cfg_dict["_time_steps"][0] == pytest.approx(0.05) and len(cfg_dict["_time_steps"]) == 20
    st = m_reset(x=HOVER, rng=jax_shim.random.PRNGKey(10), xdes=HOVER)
    assert isinstance(st, OptState) and st.yk.shape == (20, 4) and st.yk.dtype == np.float32
    st.yk.block_until_ready()
    np.testing.assert_allclose(st.yk, 0.71)
    for name in ("avg_linesearch", "stepsize", "num_steps", "grad_sqr", "avg_stepsize", "init_cost", "opt_cost"):
        float(getattr(st, name))                                         # sde_control.py:444-450,646-647
    assert float(st.stepsize) == pytest.approx(0.01)
    cfg_dict2, _, sft2, _ = load_mpc_from_cfgfile(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"), trajectory=lemniscate_state)
    s = sft2(0.01)
    assert s.shape == (13,) and s.dtype == np.float32 and hasattr(s, "block_until_ready")


def test_jax_facade_matches_call_patterns():
    f = lambda x, y=1: x + y
    g = jax_shim.jit(f).lower(1, y=2).compile()                          # sde_control.py:694,702,713
    assert g(1, y=2) == 3
    k = jax_shim.random.PRNGKey(10)
    assert k.dtype == np.uint32 and k.shape == (2,)
    a, b, c = jax_shim.random.split(k, 3)                                # sde_control.py:341
    assert a.shape == (2,) and not np.array_equal(b, c)
    np.testing.assert_array_equal(jax_shim.random.split(k, 3), jax_shim.random.split(k, 3))


def test_prefork_shape_probe_touches_no_device(monkeypatch):
    """load_single_mpc's warm-up calls (sde_control.py:706,717): the FIRST call of each compiled callable in the compiling process is a
    shape probe (right shapes, no solver created) by default; the next call in the same process is a real solve (in-process users), which
    on this GPU-less machine fails loudly inside the C ABI; SDEMPC_PREFORK=solve makes even the first call real."""
    from sde4mbrl_px4_amd.solver import SdempcError
    monkeypatch.delenv("SDEMPC_PREFORK", raising=False)
    cfg_dict, (m_reset, m_mpc), _, _ = load_mpc_from_cfgfile(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), model=synthetic_iris())
    prob = cfg_dict["_problem"]
    x0 = HOVER.copy()
    rng = jax_shim.random.PRNGKey(10)
    reset_c = jax_shim.jit(m_reset).lower(x=x0, rng=rng, xdes=x0).compile()
    st = reset_c(x=x0, rng=rng, xdes=x0)
    st.yk.block_until_ready()
    mpc_c = jax_shim.jit(m_mpc).lower(x0, rng, st, curr_t=0.01, xdes=x0).compile()
    with pytest.warns(RuntimeWarning, match="SHAPE PROBE"):               # loud: an in-process user must not mistake it for a solve
        uopt, st2, rng2, xevol = mpc_c(x0, rng, st, curr_t=0.01, xdes=x0)
    uopt.block_until_ready()
    assert np.array(uopt).shape == (20, 4) and xevol.shape == (21, 13)
    assert float(st2.num_steps) == 0.0 and np.isnan(float(st2.opt_cost)) and np.isnan(float(st2.init_cost))     # marked as unsolved
    np.testing.assert_array_equal(np.array(st2.yk), np.array(st.yk))
    assert prob._solver is None                                           # nothing touched the GPU library
    if not _has_gpu():
        with pytest.raises(SdempcError, match="no HIP device|HIP"):      # second call: the real path, no CPU fallback
            mpc_c(x0, rng, st, curr_t=0.01, xdes=x0)
        assert prob._solver is not None and not prob._solver.device_ready()
        monkeypatch.setenv("SDEMPC_PREFORK", "solve")
        mpc_d = jax_shim.jit(m_mpc).lower(x0, rng, st, curr_t=0.01, xdes=x0).compile()
        with pytest.raises(SdempcError):
            mpc_d(x0, rng, st, curr_t=0.01, xdes=x0)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_inherited_handle_is_detached_not_destroyed_after_fork():
    """A solver handle created in the parent and never used on the GPU is dropped without sdempc_destroy in a forked child (no HIP call on
    the parent's context); a fresh one is created there (sde_control.py:69-75,723-728)."""
    import multiprocessing as mp
    cfg_dict, _, _, _ = load_mpc_from_cfgfile(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), model=synthetic_iris())
    prob = cfg_dict["_problem"]
    parent = prob.solver()
    assert not parent.device_ready()

    def child(q):
        s = prob.solver()
        q.put((s is not parent, parent._h is None, s.device_ready()))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    fresh, detached, ready = q.get(timeout=60)
    p.join(30)
    assert fresh and detached and not ready and p.exitcode == 0
    assert prob.solver() is parent and parent._h is not None             # the parent keeps its own handle


def test_unavailable_model_or_trajectory_is_an_error_not_a_silent_stand_in(tmp_path, monkeypatch):
    """Every MPC YAML the reference ships points at files of the external sde4mbrl repository (iris_sitl_traj_mpc.yaml:3,6). A configured
    path that cannot be honoured raises; the synthetic vehicle / analytic lemniscate replace it only on request, with a warning."""
    import shutil
    monkeypatch.delenv("SDEMPC_ALLOW_SYNTHETIC", raising=False)
    ref_yaml = "/root/reference/launch/iris_sitl_traj_mpc.yaml"
    y = tmp_path / "traj_mpc.yaml"
    if os.path.exists(ref_yaml):
        shutil.copy(ref_yaml, y)
    else:   # GPU box: no reference tree; same two keys on top of this repo's C2 file
        y.write_text("learned_model_params: ~/nowhere/iris_sitl_sde.pkl\ntrajectory_path: ~/nowhere/fast2_lemn.csv\n"
                     + open(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml")).read())
    with pytest.raises(FileNotFoundError, match="trajectory_path|learned_model_params"):
        load_mpc_from_cfgfile(str(y))
    with pytest.raises(FileNotFoundError, match="learned_model_params"):
        load_mpc_from_cfgfile(str(y), trajectory=lemniscate_state)
    with pytest.warns(UserWarning, match="SYNTHETIC"):
        cfg_dict, _, sft, _ = load_mpc_from_cfgfile(str(y), allow_synthetic=True)
    assert sft is not None and cfg_dict["_problem"].model.num_motors == 4
    monkeypatch.setenv("SDEMPC_ALLOW_SYNTHETIC", "1")
    with pytest.warns(UserWarning):
        load_mpc_from_cfgfile(str(y))
    cfg_dict, _, _, _ = load_mpc_from_cfgfile(str(y), model=synthetic_iris(), trajectory=lemniscate_state)   # explicit: no warning needed
    assert cfg_dict["_problem"].convert_to_enu is True


def test_frame_contract_hold_target_equals_solver_state():
    """SPEC.md §1a: x arrives NED, targets are ENU, convert_to_enu=True flips x. In hold mode the node passes xdes = enu2ned(x)
    (sde_control.py:400): the reference window then starts exactly at the solver's initial state (zero error at t0)."""
    from sde4mbrl_px4_amd.utils import enu2ned
    from sde4mbrl_px4_amd.workload import random_initial_states
    cfg_dict, _, _, _ = load_mpc_from_cfgfile(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), model=synthetic_iris())
    prob = cfg_dict["_problem"]
    x_ned = random_initial_states(1, 7)[0]
    xdes = enu2ned(x_ned, np)
    xs = enu2ned(x_ned, np) if prob.convert_to_enu else x_ned
    xref = prob.xref(0.0, xdes)
    assert xref.shape == (21, 13) and np.array_equal(xref[0], xs) and np.array_equal(xref[-1], xs)
    assert not np.array_equal(xs[:3], x_ned[:3])                                       # the flip is not the identity
    np.testing.assert_allclose(enu2ned(xs, np), x_ned, atol=2e-7)                      # involution up to rounding
    assert xs[2] == -x_ned[2] and xs[0] == x_ned[1] and xs[11] == -x_ned[11]


def test_handle_options_host_side():
    """sdempc_set_option / sdempc_get_option (include/sdempc.h): defaults, environment defaults read at create, validation; host-only."""
    from sde4mbrl_px4_amd import load_mpc_config
    from sde4mbrl_px4_amd.solver import SdeMpcSolver, SdempcError
    cfg = load_mpc_config(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"))
    for k in ("SDEMPC_LANE", "SDEMPC_COOP", "SDEMPC_SPEC", "SDEMPC_PK", "SDEMPC_USTG", "SDEMPC_COOP_LAUNCH", "SDEMPC_COOP_FENCE", "SDEMPC_COOP_SPIN_US", "SDEMPC_HEX", "SDEMPC_DUO"):
        os.environ.pop(k, None)
    S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=2)
    assert [S.get_option(k) for k in ("lane", "coop", "spec", "pk", "ustg", "coop_launch", "coop_fence")] == [1, 1, 1, -1, -1, 0, 0]
    assert S.get_option("hex") == 1 and S.get_option("duo") == -1          # six-team workgroups for launches that fill the device: on
    S.set_option("hex", 0); assert S.get_option("hex") == 0; S.set_option("hex", 1)
    with pytest.raises(SdempcError):
        S.set_option("hex", 2)
    assert S.get_option("coop_spin_us") == 100_000                       # derived budget before the first cooperative solve: 100 ms
    S.set_option("coop_spin_us", 2500); assert S.get_option("coop_spin_us") == 2500
    S.set_option("pk", 0); S.set_option("ustg", 1); S.set_option("coop", 0); S.set_option("coop_fence", 1)
    assert (S.get_option("pk"), S.get_option("ustg"), S.get_option("coop"), S.get_option("coop_fence")) == (0, 1, 0, 1)
    for key, bad in (("lane", 2), ("pk", 3), ("coop_spin_us", -2), ("spec", -1)):
        with pytest.raises(SdempcError):
            S.set_option(key, bad)
    with pytest.raises(SdempcError):
        S._check(S.lib.sdempc_set_option(S._h, 99, 0))
    assert not S.device_ready()
    S.close()
    os.environ["SDEMPC_COOP"] = "0"; os.environ["SDEMPC_COOP_SPIN_US"] = "777"
    try:
        S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=1)
        assert S.get_option("coop") == 0 and S.get_option("coop_spin_us") == 777
        os.environ["SDEMPC_COOP"] = "1"                                  # read once at create: later changes do not reach the handle
        assert S.get_option("coop") == 0
        S.close()
        S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=1, options={"coop": 1})
        assert S.get_option("coop") == 1
        S.close()
    finally:
        os.environ.pop("SDEMPC_COOP", None); os.environ.pop("SDEMPC_COOP_SPIN_US", None)


def test_shared_block_layouts_and_command_selection():
    _, (m_reset, _), _, _ = load_mpc_from_cfgfile(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"))
    sh = SharedBlocks.create(50, 20, 4, m_reset())
    assert sh.curr_state.dtype == np.float32 and sh.curr_state.shape == (13,)
    assert sh.u_opt.shape == (50, 4) and sh.u_opt.dtype == np.float32
    assert sh.w_opt.shape == (50, 4) and sh.w_opt.dtype == np.float64
    assert sh.info_mpc_pre.dtype == np.float64 and sh.info_mpc_pre.shape == (3,)
    assert sh.opt_info.dtype == np.float32 and sh.opt_info.shape == (9,) and sh.opt_info[0] == -1.0
    u = np.arange(80, dtype=np.float32).reshape(20, 4)
    w = np.arange(80, dtype=np.float64).reshape(20, 4)
    assert select_command(1000.0, -1.0, 50000.0, u, w, 20) is None       # no solution yet (sde_control.py:284-288)
    idx, mot, wo = select_command(1_120_000.0, 1_000_000.0, 50000.0, u, w, 20)
    assert idx == 2 and mot.shape == (6,) and list(mot[:4]) == [8, 9, 10, 11] and list(mot[4:]) == [0, 0]
    idx, _, _ = select_command(9_000_000.0, 1_000_000.0, 50000.0, u, w, 20)
    assert idx == 19                                                     # clamped to the last row (sde_control.py:294-298)
    assert CONTROL_STATE == {"none": 0, "reset": 1, "test": 2, "pos": 3, "idle": 4, "traj": 5}


# ---- N3: importer for pickled parameter trees (layout supplied by the caller; nothing about the external format is guessed) ----
def test_param_tree_importer_round_trip(tmp_path):
    import pickle

    import yaml

    from sde4mbrl_px4_amd import synthetic_hexa
    from sde4mbrl_px4_amd.importer import import_sde_pickle, load_param_tree

    ref = synthetic_hexa()
    m = ref.num_motors
    W1 = np.concatenate([ref.W1z[:32], ref.W1u], axis=1)            # [32, 6+m]
    tree = {   # Haiku-style: module -> {w [in, out], b}; arrays are plain numpy (jax.device_get before pickling)
        "drift/~/linear_0": {"w": W1.T.copy(), "b": ref.b1[:32].copy()},
        "drift/~/linear_1": {"w": ref.W2.T.copy(), "b": ref.b2.copy()},
        "drift/~/linear_2": {"w": ref.W3.T.copy(), "b": ref.b3.copy()},
        "density": {"linear_0": {"w": ref.W1z[32:].T.copy(), "b": ref.b1[32:].copy()}, "linear_1": {"w": ref.w3n[:, None].copy(), "b": np.array([ref.b3n], np.float32)}},
        "prior": {"mass": np.float32(ref.mass), "inertia": ref.inertia},
    }
    pk = tmp_path / "hexa_sde.pkl"
    pk.write_bytes(pickle.dumps(tree))
    mapping = {
        "drift_l1": {"w": "drift/~/linear_0/w", "b": "drift/~/linear_0/b"},
        "drift_l2": {"w": "drift/~/linear_1/w", "b": "drift/~/linear_1/b"},
        "drift_out": {"w": "drift/~/linear_2/w", "b": "drift/~/linear_2/b"},
        "density_l1": {"w": "density/linear_0/w", "b": "density/linear_0/b"},
        "density_out": {"w": "density/linear_1/w", "b": "density/linear_1/b"},
        "physics": {"mass": "prior/mass", "inertia": "prior/inertia", "grav": ref.grav, "thrust_poly": ref.thrust_poly.tolist(),
                    "moment_poly": ref.moment_poly.tolist(), "rotor_x": ref.rotor_x.tolist(), "rotor_y": ref.rotor_y.tolist(),
                    "rotor_dir": ref.rotor_dir.tolist(), "res_force_scale": ref.res_force_scale.tolist(),
                    "res_torque_scale": ref.res_torque_scale.tolist(), "sigma": ref.sigma.tolist()},
    }
    mp = tmp_path / "mapping.yaml"
    mp.write_text(yaml.safe_dump(mapping))
    got = import_sde_pickle(str(pk), str(mp))
    assert got.num_motors == m and got.to_blob() == ref.to_blob()
    # wrong orientation / architecture is reported with the shape
    bad = dict(mapping, drift_l2={"w": "drift/~/linear_0/w", "b": "drift/~/linear_1/b"})
    with pytest.raises(ValueError, match="drift_l2"):
        import_sde_pickle(str(pk), bad)
    with pytest.raises(KeyError, match="no leaf"):
        import_sde_pickle(str(pk), dict(mapping, drift_out={"w": "nope/w", "b": "drift/~/linear_2/b"}))

    # a pickle that names any other callable is refused before it can run
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned",))
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        load_param_tree(pickle.dumps({"w": Evil()}))
