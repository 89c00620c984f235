"""Drop-in for `sde4mbrlExamples.rotor_uav.sde_mpc_design` (import site sde_control.py:12).

`load_mpc_from_cfgfile(mpc_dir, convert_to_enu=True)` returns
    cfg_dict, (m_reset, m_mpc), state_from_traj, None
with the call signatures the reference node uses:
    m_reset(x=, rng=, xdes=) -> opt_state                                     sde_control.py:702,706,345-346,389-394
    m_mpc(x, rng, opt_state, curr_t=, xdes=) -> (uopt, opt_state, rng, xevol)  sde_control.py:713,717,349-350,400-416
    state_from_traj(t) -> f32[13]  (None for position-control YAMLs)           sde_control.py:164,177,206,694
    cfg_dict['_time_steps'][0] = dt                                            sde_control.py:167,174
Returned arrays are numpy arrays wrapped so that `.block_until_ready()` exists (sde_control.py:420,707,718).
opt_state exposes yk and the seven telemetry scalars read at sde_control.py:444-450,646-647.

Frame contract (SPEC.md §1a). The node hands over the vehicle state `x` as it arrives in the MPC_FULL_STATE message, documented as
NED (sde_control.py:228-232,246), while every target it builds is ENU: the position set-point (:186-192), `state_from_traj` (:206,
trajectory CSVs are ENU), and the hold target `enu2ned(curr_state, np)` (:400), i.e. the NED state flipped to ENU. The factory is
called with `convert_to_enu=True` (:685): the solver converts `x` itself. So, with convert_to_enu=True, m_mpc maps x -> enu2ned(x)
(the flip is an involution), solves in ENU / FLU (the frame of SPEC.md §5), takes `xdes` / `state_from_traj` as ENU, and returns
`xevol` flipped back to the frame of `x` (its body rates, rows 10..12, go to the FCU: sde_control.py:432). `uopt` is frame-free.
With convert_to_enu=False nothing is converted: x, xdes and the trajectory are taken to be in the solver's frame already.

The solve itself runs in the HIP kernels behind include/sdempc.h; this module is host glue only.
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass, field
from typing import Callable, NamedTuple, Optional

import numpy as np

from . import prng, workload
from .config import MPCConfig, load_mpc_config
from .model import RotorSDEModel, synthetic_hexa, synthetic_iris
from .utils import TrajectoryCSV, enu2ned


class DeviceArray(np.ndarray):
    """numpy array with the one JAX-array method the reference calls."""

    def block_until_ready(self):
        return self


def _arr(a) -> DeviceArray:
    return np.ascontiguousarray(a, dtype=np.float32).view(DeviceArray)


class OptState(NamedTuple):
    """Optimiser state threaded through successive m_mpc calls (sde_control.py:345,400-416,444-450)."""
    yk: DeviceArray                # [H, m] warm start for the next solve
    avg_linesearch: np.float32
    stepsize: np.float32
    num_steps: np.float32
    grad_sqr: np.float32
    avg_stepsize: np.float32
    init_cost: np.float32
    opt_cost: np.float32
    num_ls_trials: np.float32 = np.float32(0.0)


def _next_key(rng):
    """SPEC.md §7.3: `new_rng, sub = split(rng)` with JAX's threefry2x32 split; the solve draws its noise tensor as
    normal(sub, (P, H, 6)) on the device. Returns (new_rng, sub), both uint32[2]."""
    k = prng.split(np.asarray(rng, dtype=np.uint32).reshape(2), 2)
    return k[0].copy(), k[1].copy()


@dataclass
class MpcProblem:
    """One loaded MPC YAML: config, model, solver handle (created lazily) and reference source."""
    cfg: MPCConfig
    model: RotorSDEModel
    state_from_traj: Optional[Callable] = None
    shift_warm_start: bool = True
    convert_to_enu: bool = True           # sde_control.py:685; see the frame contract in the module docstring
    _solver: object = field(default=None, repr=False)
    _pid: int = field(default=-1, repr=False)

    def solver(self):
        # HIP contexts do not survive fork(): the reference builds its solvers in the parent and uses them in the forked mpc_process
        # (sde_control.py:69-75,723-728), so the handle is (re)created per process. A handle inherited through fork() is never
        # touched again: no HIP call may run on the parent's context in the child, not even the frees of sdempc_destroy, so it is
        # detached (leaked). If the parent had already initialised the GPU through it (a real solve before the fork) the child's HIP
        # runtime is unusable: fail loudly instead of hanging inside the control process.
        if self._solver is not None and self._pid != os.getpid():
            inherited, self._solver = self._solver, None
            was_ready = inherited.device_ready()
            inherited.detach()
            if was_ready:
                raise RuntimeError(
                    "sde4mbrl_px4_amd: this process was forked after its parent had already run a solve on the GPU (HIP state does "
                    "not survive fork()). Keep real solver calls out of the parent: the first call of each compiled callable there is "
                    "answered by a shape probe (jax_shim, SDEMPC_PREFORK), or start the worker with the 'spawn' method.")
        if self._solver is None:
            from .solver import SdeMpcSolver
            self._solver = SdeMpcSolver(self.cfg, self.model, max_batch=1)
            self._pid = os.getpid()
        return self._solver

    def xref(self, curr_t: float, xdes) -> np.ndarray:
        if self.state_from_traj is not None:
            return workload.reference_window(float(curr_t), self.cfg.time_steps, self.state_from_traj)
        return workload.constant_reference(np.asarray(xdes, np.float32), self.cfg.horizon)

    # ---- the two callables ------------------------------------------------------------------------
    def m_reset(self, x=None, rng=None, xdes=None) -> OptState:
        H, m = self.cfg.horizon, self.cfg.num_motors
        yk = np.tile(np.asarray(self.cfg.uref, np.float32), (H, 1))
        s0 = self.cfg.ls_init_stepsize if self.cfg.ls_maxls > 0 else self.cfg.stepsize
        z = np.float32(0.0)
        return OptState(_arr(yk), z, np.float32(s0), z, z, z, z, z, z)

    def shape_probe_m_mpc(self, x, rng, opt_state: OptState, curr_t=0.0, xdes=None):
        """Outputs with the shapes/dtypes of m_mpc and no device work (see jax_shim._Compiled): the warm start as uopt,
        the current state repeated as xevol. Used only for the reference's pre-fork warm-up calls."""
        x = np.asarray(x, np.float32).reshape(13)
        nan = np.float32(np.nan)       # marks the state as "not the result of a solve": num_steps 0, costs NaN (jax_shim warns as well)
        marked = opt_state._replace(num_steps=np.float32(0.0), init_cost=nan, opt_cost=nan)
        return _arr(np.asarray(opt_state.yk, np.float32)), marked, rng, _arr(np.tile(x, (self.cfg.horizon + 1, 1)))

    def m_mpc(self, x, rng, opt_state: OptState, curr_t=0.0, xdes=None):
        x = np.asarray(x, np.float32).reshape(13)
        xs = enu2ned(x, np) if self.convert_to_enu else x           # vehicle state in the solver's frame (module docstring)
        xdes = xs if xdes is None else np.asarray(xdes, np.float32).reshape(13)
        new_rng, sub = _next_key(rng)
        xref = self.xref(float(curr_t), xdes)[None]
        u0 = np.asarray(opt_state.yk, np.float32)[None]
        uopt, xevol, info = self.solver().solve_keys(xs[None], xref, sub[None], u0, np.array([opt_state.stepsize], np.float32))
        if self.convert_to_enu:                                     # predicted states back in the frame of x
            xevol = enu2ned(xevol, np)
        uo = uopt[0]
        yk = np.concatenate([uo[1:], uo[-1:]], axis=0) if self.shift_warm_start else uo
        i = info[0]
        st = OptState(_arr(yk), np.float32(i[0]), np.float32(i[1]), np.float32(i[2]), np.float32(i[3]), np.float32(i[4]),
                      np.float32(i[5]), np.float32(i[6]), np.float32(i[7]))
        return _arr(uo), st, new_rng, _arr(xevol[0])


def _allow_synthetic(flag) -> bool:
    return bool(flag) if flag is not None else os.environ.get("SDEMPC_ALLOW_SYNTHETIC") == "1"


def _pick_model(cfg: MPCConfig, model, allow_synthetic=None) -> RotorSDEModel:
    """The vehicle model: `model` if given; else what `learned_model_params` names (iris_sitl_traj_mpc.yaml:3). A configured file that
    cannot be honoured raises — the synthetic vehicle replaces it only on request (allow_synthetic=True / SDEMPC_ALLOW_SYNTHETIC=1),
    and never silently."""
    if model is not None:
        return model
    lm = os.path.expanduser(cfg.learned_model_params) if cfg.learned_model_params else None
    why = None
    if lm is None:
        why = "the MPC YAML names no learned_model_params"
    elif not os.path.exists(lm):
        why = f"learned_model_params '{lm}' does not exist (the reference's pickles live in the external sde4mbrl repository)"
    elif lm.endswith(".npz"):
        m = RotorSDEModel.load_npz(lm)
        if m.num_motors != cfg.num_motors:
            raise ValueError(f"{lm}: model has {m.num_motors} motors, config has {cfg.num_motors}")
        return m
    elif lm.endswith(".pkl"):
        # a pickle of the external sde4mbrl repo: its layout is not in the reference, so it is read only when the user states the
        # layout in <name>.mapping.yaml beside it (importer.py)
        mp = lm[:-4] + ".mapping.yaml"
        if os.path.exists(mp):
            from .importer import import_sde_pickle
            mdl = import_sde_pickle(lm, mp)
            if mdl.num_motors != cfg.num_motors:
                raise ValueError(f"{lm}: model has {mdl.num_motors} motors, config has {cfg.num_motors}")
            return mdl
        why = f"'{lm}' is a pickle without a layout description '{mp}' (importer.py)"
    else:
        why = f"learned_model_params '{lm}': unknown file type (expected .npz, or .pkl with a .mapping.yaml)"
    if lm is not None and not _allow_synthetic(allow_synthetic):
        raise FileNotFoundError(f"sde4mbrl_px4_amd: {why}. Pass model=..., fix the path, or opt in to the synthetic stand-in vehicle "
                                "with allow_synthetic=True / SDEMPC_ALLOW_SYNTHETIC=1.")
    warnings.warn(f"sde4mbrl_px4_amd: {why}: using this build's SYNTHETIC {'iris' if cfg.num_motors == 4 else 'hexa'} vehicle "
                  "(model.py; random residual weights, not a learned model)", stacklevel=3)
    return synthetic_iris() if cfg.num_motors == 4 else synthetic_hexa()


def load_mpc_problem(mpc_dir: str, convert_to_enu: bool = True, model=None, horizon=None, num_particles=None,
                     trajectory=None, overrides=None, allow_synthetic=None) -> MpcProblem:
    cfg = load_mpc_config(mpc_dir)
    over = dict(overrides or {})
    if horizon is not None:
        over.update(horizon=int(horizon), num_short_dt=int(horizon))
    if num_particles is not None:
        over.update(num_particles=int(num_particles))
    if over:
        cfg = cfg.replace(**over)
    sft = None
    if trajectory is not None:
        sft = trajectory
    elif cfg.trajectory_path:
        path = os.path.expanduser(cfg.trajectory_path)
        if os.path.exists(path):
            sft = TrajectoryCSV(path, ned=not convert_to_enu)
        elif _allow_synthetic(allow_synthetic):
            # the shipped YAMLs point at CSVs of the external repo: analytic lemniscate instead, on request only
            warnings.warn(f"sde4mbrl_px4_amd: trajectory_path '{path}' does not exist: using the analytic lemniscate stand-in", stacklevel=2)
            sft = workload.lemniscate_state
        else:
            raise FileNotFoundError(f"sde4mbrl_px4_amd: trajectory_path '{path}' does not exist. Pass trajectory=..., fix the path, or opt "
                                    "in to the analytic lemniscate with allow_synthetic=True / SDEMPC_ALLOW_SYNTHETIC=1.")
    return MpcProblem(cfg=cfg, model=_pick_model(cfg, model, allow_synthetic), state_from_traj=sft, convert_to_enu=bool(convert_to_enu))


def load_mpc_from_cfgfile(mpc_dir: str, convert_to_enu: bool = True, **kw):
    """Reference-shaped factory (sde_control.py:685)."""
    prob = load_mpc_problem(mpc_dir, convert_to_enu=convert_to_enu, **kw)
    cfg_dict = {"_time_steps": prob.cfg.time_steps, "horizon": prob.cfg.horizon, "num_particles": prob.cfg.num_particles,
                "cost_params": {"uref": list(prob.cfg.uref)}, "_problem": prob}
    sft = None
    if prob.state_from_traj is not None:
        def sft(t, _f=prob.state_from_traj):
            return _arr(np.asarray(_f(np.float64(t)), np.float32).reshape(13))
    return cfg_dict, (prob.m_reset, prob.m_mpc), sft, None
