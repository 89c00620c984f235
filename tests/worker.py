"""ROS-free replay of the reference's MPC worker loop and of the command-selection rule.

`MpcWorker.step()` reproduces one iteration of `SDEControlROS.mpc_process_fn`
(sde4mbrl_px4/mpc_controller/sde_control.py:365-450): mode-dependent resets (:387-396), exactly one
position/trajectory solve per mode (:398-416, idle alternation :406-408), the A8 post-processing
(:428-432) and the shared-block writes (:437-450). `SharedBlocks` has the layouts of
`multi_process_shared_variables` (:616-656) as plain numpy arrays; `select_command` is the index rule
of `mpc_state_callback` (:283-306). ROS, MAVLink and the process/lock plumbing are out of scope.

TEST HARNESS (SURVEY.md §8f N1, §8c last row): a statement-by-statement counterpart of the reference's caller, kept next to tests/replay.py —
not a component of the product package (it drives the product through the reference's own entry points: load_mpc_from_cfgfile, m_reset, m_mpc).
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import numpy as np

from sde4mbrl_px4_amd import jax_shim
from sde4mbrl_px4_amd.sde_mpc_design import MpcProblem
from sde4mbrl_px4_amd.utils import enu2ned
from sde4mbrl_px4_amd.workload import HOVER

CONTROL_STATE = {"none": 0, "reset": 1, "test": 2, "pos": 3, "idle": 4, "traj": 5}          # sde_control.py:46
KEY2INDEX_PRE = {"sample_time_prempc": 0, "duration": 1, "ctrl_state": 2}                     # sde_control.py:639
KEY2INDEX_INFO = {"sample_time_posmpc": 0, "avg_linesearch": 1, "stepsize": 2, "num_steps": 3, "grad_norm": 4,
                  "avg_stepsize": 5, "cost0": 6, "costT": 7, "solveTime": 8}                  # sde_control.py:648-649


@dataclass
class SharedBlocks:
    """Arrays with the shapes/dtypes of the reference's six shared-memory blocks (sde_control.py:620-656)."""
    curr_state: np.ndarray       # f32[13]
    u_opt: np.ndarray            # f32[max(H_traj,H_pos), m]
    w_opt: np.ndarray            # f64[H, 4]   thrust + desired body rates
    info_mpc_pre: np.ndarray     # f64[3]      sample_time us, duration s, ctrl_state
    opt_info: np.ndarray         # f32[9]
    target_setpoint: np.ndarray  # f32[13]

    @staticmethod
    def create(h_traj: int, h_pos: int, m: int, default_opt_state) -> "SharedBlocks":
        hmax = max(h_traj, h_pos)
        o = default_opt_state
        info = np.array([-1.0, o.avg_linesearch, o.stepsize, o.num_steps, o.grad_sqr, o.avg_stepsize, o.init_cost, o.opt_cost, 0.0], dtype=np.float32)
        return SharedBlocks(HOVER.copy(), np.zeros((hmax, m), np.float32), np.zeros((hmax, 4), np.float64),
                            np.array([0.0, 0.0, CONTROL_STATE["none"]]), info, HOVER.copy())


def select_command(sample_time_usec, tsample_mpc_usec, dt_usec, u_opt, w_opt, horizon):
    """Time-aligned row of the latest solution (sde_control.py:283-306). Returns (index, motors[6], wopt[4]) or None."""
    if tsample_mpc_usec <= 0:
        return None
    idx = int((sample_time_usec - tsample_mpc_usec) / dt_usec)
    if idx >= horizon:
        idx = horizon - 1
    u = np.asarray(u_opt[idx, :])
    if u.shape[0] < 6:
        u = np.concatenate((u, np.zeros((6 - u.shape[0],))))
    return idx, u, np.asarray(w_opt[idx, :])


class MpcWorker:
    def __init__(self, traj: MpcProblem, pos: MpcProblem, seed: int = 0):
        assert traj.state_from_traj is not None and pos.state_from_traj is None      # sde_control.py:164,177
        self.traj, self.pos = traj, pos
        rng_ctrl = jax_shim.random.PRNGKey(seed)                                        # :338
        _, self.rng_traj, self.rng_pos = jax_shim.random.split(rng_ctrl, 3)            # :341
        x0 = HOVER.copy()
        self.opt_state_traj = traj.m_reset(x=x0, rng=self.rng_traj, xdes=x0)           # :345
        self.opt_state_pos = pos.m_reset(x=x0, rng=self.rng_pos, xdes=x0)              # :346
        self.shared = SharedBlocks.create(traj.cfg.horizon, pos.cfg.horizon, traj.cfg.num_motors, self.opt_state_traj)
        self._curr_ctrl = None
        self._idle_traj = False
        self.dt_usec_traj = float(traj.cfg.time_steps[0]) * 1e6                        # :167
        self.dt_usec_pos = float(pos.cfg.time_steps[0]) * 1e6                          # :174

    def warm_up(self):
        x0 = HOVER.copy()
        self.traj.m_mpc(x0, self.rng_traj, self.opt_state_traj, curr_t=0.0, xdes=x0)   # :349
        self.pos.m_mpc(x0, self.rng_pos, self.opt_state_pos, curr_t=0.0, xdes=x0)      # :350

    def step(self, curr_state, ctrl_state: int, duration: float, target_x, sample_time_usec: float):
        curr_state = np.asarray(curr_state, np.float32)
        cs = CONTROL_STATE
        t0 = time.time()
        if self._curr_ctrl is None or (self._curr_ctrl == "none" and ctrl_state != cs["none"]):          # :387-390
            self.opt_state_traj = self.traj.m_reset(x=curr_state, rng=self.rng_traj, xdes=curr_state)
            self.opt_state_pos = self.pos.m_reset(x=curr_state, rng=self.rng_pos, xdes=curr_state)
        if ctrl_state == cs["idle"] and self._curr_ctrl in (None, "none", "pos"):                       # :392-396
            self.opt_state_traj = self.traj.m_reset(x=curr_state, rng=self.rng_traj, xdes=curr_state)
            self._curr_ctrl = "idle"
            self._idle_traj = True
        if ctrl_state == cs["none"]:                                                                    # :398-400
            self._curr_ctrl = "none"
            uopt, self.opt_state_pos, self.rng_pos, evol = self.pos.m_mpc(curr_state, self.rng_pos, self.opt_state_pos, curr_t=0.0,
                                                                        xdes=enu2ned(curr_state, np))
        elif ctrl_state == cs["idle"]:                                                                  # :402-408
            self._curr_ctrl = "idle"
            uopt, self.opt_state_pos, self.rng_pos, evol = self.pos.m_mpc(curr_state, self.rng_pos, self.opt_state_pos, curr_t=0.0, xdes=target_x)
            self._idle_traj = not self._idle_traj
            if self._idle_traj:
                _, self.opt_state_traj, self.rng_traj, _ = self.traj.m_mpc(curr_state, self.rng_traj, self.opt_state_traj, curr_t=duration, xdes=curr_state)
        elif ctrl_state == cs["traj"]:                                                                  # :410-412
            self._curr_ctrl = "traj"
            uopt, self.opt_state_traj, self.rng_traj, evol = self.traj.m_mpc(curr_state, self.rng_traj, self.opt_state_traj, curr_t=duration, xdes=curr_state)
        elif ctrl_state == cs["pos"]:                                                                   # :414-416
            self._curr_ctrl = "pos"
            uopt, self.opt_state_pos, self.rng_pos, evol = self.pos.m_mpc(curr_state, self.rng_pos, self.opt_state_pos, curr_t=0.0, xdes=target_x)
        else:
            raise ValueError(f"Unknown control state: {ctrl_state}")                                    # :419
        uopt.block_until_ready()                                                                        # :420
        solve_time = time.time() - t0
        uopt = np.array(uopt)                                                                           # :428
        thrust = np.sum(uopt, axis=1) / uopt.shape[1]                                                   # :431
        wopt = np.array([thrust, evol[1:, 10], evol[1:, 11], evol[1:, 12]]).T                           # :432
        st = self.opt_state_traj if self._curr_ctrl in ("traj", "idle") else self.opt_state_pos        # :435
        sh, k = self.shared, KEY2INDEX_INFO
        sh.u_opt[:uopt.shape[0], :] = uopt                                                              # :439
        sh.w_opt[:wopt.shape[0], :] = wopt                                                              # :441
        sh.opt_info[k["sample_time_posmpc"]] = sample_time_usec                                         # :442 (usec squeezed into f32, as in the reference)
        sh.opt_info[k["solveTime"]] = solve_time
        sh.opt_info[k["avg_linesearch"]] = float(st.avg_linesearch)
        sh.opt_info[k["stepsize"]] = float(st.stepsize)
        sh.opt_info[k["num_steps"]] = float(st.num_steps)
        sh.opt_info[k["grad_norm"]] = float(st.grad_sqr)
        sh.opt_info[k["avg_stepsize"]] = float(st.avg_stepsize)
        sh.opt_info[k["cost0"]] = float(st.init_cost)
        sh.opt_info[k["costT"]] = float(st.opt_cost)
        return uopt, wopt, st
