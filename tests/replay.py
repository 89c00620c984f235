"""Replay of the reference's per-tick driver: one scripted mode sequence fed to `MpcWorker.step`
(= one iteration of `mpc_process_fn`, sde_control.py:365-450) and to `select_command` (= the index rule of `mpc_state_callback`,
sde_control.py:283-306). Shared by tests/golden/make_worker_replay.py (oracle-backed worker -> committed fixture), the CPU test that pins
that fixture and the GPU test that replays the same sequence through the HIP path. Test infrastructure."""
import os

import numpy as np

import orc
from cases import CDIR, GOLDEN_DIR
from sde4mbrl_px4_amd import synthetic_iris
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.sde_mpc_design import MpcProblem, load_mpc_problem
from sde4mbrl_px4_amd.utils import enu2ned
from worker import CONTROL_STATE, MpcWorker, select_command

FIXTURE = os.path.join(GOLDEN_DIR, "worker_replay.npz")
SMALL = dict(max_iter=6, max_no_improvement_iter=6)
# none -> pos -> idle x k -> traj x k -> pos -> none -> idle -> traj: every transition of sde_control.py:387-416, idle alternation included
MODES = ["none"] * 3 + ["pos"] * 4 + ["idle"] * 6 + ["traj"] * 8 + ["pos"] * 4 + ["none"] * 2 + ["idle"] * 3 + ["traj"] * 4
LAGS_US = (0.0, 30_000.0, 130_000.0, 5_000_000.0)      # how late the state callback reads the solution (sde_control.py:292-298)


class _OracleSolver:
    """Stands in for SdeMpcSolver inside an MpcProblem: same solve_keys signature, computed by the CPU oracle."""

    def __init__(self, cfg, model):
        self.cfg, self.O = cfg, orc.Oracle(cfg, model)

    def solve_keys(self, x0, xref, keys, u_init, stepsize_in):
        B = x0.shape[0]
        out = [self.O.solve(x0[b], xref[b], orc.noise_from_key(keys[b], self.cfg.num_particles, self.cfg.horizon), u_init[b], float(stepsize_in[b]))[:3]
               for b in range(B)]
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out])


class OracleProblem(MpcProblem):
    def solver(self):
        if self._solver is None:
            self._solver = _OracleSolver(self.cfg, self.model)
        return self._solver


def problems(oracle: bool):
    model = synthetic_iris()
    traj = load_mpc_problem(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"), horizon=12, num_particles=40, trajectory=W.lemniscate_state, model=model, overrides=SMALL)
    pos = load_mpc_problem(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), horizon=10, model=model, overrides=SMALL)
    if oracle:
        traj = OracleProblem(cfg=traj.cfg, model=traj.model, state_from_traj=traj.state_from_traj, convert_to_enu=traj.convert_to_enu)
        pos = OracleProblem(cfg=pos.cfg, model=pos.model, state_from_traj=None, convert_to_enu=pos.convert_to_enu)
    return traj, pos


def inputs():
    """Per tick: vehicle state as the FCU reports it (NED), mode, trajectory clock, ENU target, sample time (us)."""
    rng = np.random.default_rng(2024)
    n = len(MODES)
    x = W.random_initial_states(1, 40)[0]
    states, durs, targets, tus = [], [], [], []
    t_us, dur = 1_000_000.0, 0.0
    target = W.HOVER.copy(); target[0:3] = [0.4, -0.3, 1.2]
    for k, mode in enumerate(MODES):
        x = x.copy()
        x[0:6] += np.float32(0.02) * rng.standard_normal(6).astype(np.float32)
        x[10:13] += np.float32(0.01) * rng.standard_normal(3).astype(np.float32)
        q = x[6:10] + np.float32(0.01) * rng.standard_normal(4).astype(np.float32)
        x[6:10] = q / np.linalg.norm(q)
        dur = dur + 0.02 if mode == "traj" else (0.0 if mode != "idle" else dur)
        tgt = np.asarray(W.lemniscate_state(0.0), np.float32) if mode == "idle" else target      # idle: state_from_traj(0.0) (sde_control.py:206)
        states.append(x.astype(np.float32)); durs.append(dur); targets.append(tgt.astype(np.float32)); tus.append(t_us)
        t_us += 20_000.0
    return np.stack(states), np.asarray(durs), np.stack(targets), np.asarray(tus)


def run(oracle: bool):
    traj, pos = problems(oracle)
    wk = MpcWorker(traj, pos, seed=10)                       # launch seed (iris_sdectrl.launch:8)
    states, durs, targets, tus = inputs()
    H = max(traj.cfg.horizon, pos.cfg.horizon)
    n = len(MODES)
    rec = dict(u_opt=np.zeros((n, H, 4), np.float32), w_opt=np.zeros((n, H, 4), np.float64), opt_info=np.zeros((n, 7), np.float32),
               sel_idx=np.zeros((n, len(LAGS_US)), np.int32), sel_u=np.zeros((n, len(LAGS_US), 6), np.float64), sel_w=np.zeros((n, len(LAGS_US), 4), np.float64))
    for k, mode in enumerate(MODES):
        wk.step(states[k], CONTROL_STATE[mode], float(durs[k]), targets[k], float(tus[k]))
        sh = wk.shared
        rec["u_opt"][k], rec["w_opt"][k], rec["opt_info"][k] = sh.u_opt, sh.w_opt, sh.opt_info[1:8]
        hz = traj.cfg.horizon if mode == "traj" else pos.cfg.horizon          # the callback's _dt_usec / horizon follow the mode (:186-219)
        dt_us = wk.dt_usec_traj if mode == "traj" else wk.dt_usec_pos
        for j, lag in enumerate(LAGS_US):
            idx, u6, w4 = select_command(tus[k] + lag, tus[k], dt_us, sh.u_opt, sh.w_opt, hz)
            rec["sel_idx"][k, j], rec["sel_u"][k, j], rec["sel_w"][k, j] = idx, u6, w4
    rec.update(states=states, durations=durs, targets=targets, sample_time_us=tus, modes=np.array([CONTROL_STATE[m] for m in MODES], np.int32))
    return rec
