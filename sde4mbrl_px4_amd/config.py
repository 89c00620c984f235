"""MPC hyper-parameter schema: parses the reference's MPC YAML files verbatim.

Schema source: launch/iris_sitl_traj_mpc.yaml:1-85, launch/iris_sitl_posctrl_mpc.yaml:1-101,
launch/hexa_sitl_traj_mpc.yaml:1-64 (and the three sibling files) of wuwushrek/sde4mbrl_px4.
Only `horizon` / `num_particles` are overridden by the benchmark configs.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import yaml

from ._abi import MAX_MOTORS, SdempcCfg


def _vec3(v):
    a = np.atleast_1d(np.asarray(v, dtype=np.float32))
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise ValueError(f"expected scalar or 3-vector, got {v!r}")
    return [float(x) for x in a]


@dataclass
class MPCConfig:
    """One MPC YAML file. Field names follow the YAML keys."""
    learned_model_params: Optional[str] = None
    trajectory_path: Optional[str] = None
    input_id: List[int] = field(default_factory=lambda: [0, 1, 2, 3])
    input_bound: List[List[float]] = field(default_factory=lambda: [[1e-4, 1.0]] * 4)
    enforce_ubound: bool = True
    # cost_params
    uref: List[float] = field(default_factory=lambda: [0.71] * 4)
    uerr: float = 1.0
    perr: List[float] = field(default_factory=lambda: [100.0, 100.0, 200.0])
    verr: List[float] = field(default_factory=lambda: [5.0, 5.0, 10.0])
    qerr: List[float] = field(default_factory=lambda: [1.0, 1.0, 100.0])
    werr: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0])
    res_mult: float = 0.01
    u_slew_coeff: float = 0.0
    u_slew_constr: Optional[List[List[float]]] = None
    u_slew_constr_coeff: float = 0.0
    # state_constr (iris_sitl_traj_mpc.yaml:16-29, commented out in every shipped YAML): penalty form only (slack_proximal: False)
    state_id: Optional[List[int]] = None
    state_penalty: Optional[List[float]] = None
    state_bound: Optional[List[List[float]]] = None
    constr_pen: float = 1.0
    # horizon / time grid
    horizon: int = 20
    num_short_dt: int = 20
    short_step_dt: float = 0.05
    long_step_dt: float = 0.05
    discount: float = 1.0
    num_particles: int = 1
    # apg_mpc
    stepsize: float = 1.0
    max_iter: int = 200
    max_no_improvement_iter: int = 200
    moment_scale: Optional[float] = None
    beta_init: float = 0.25
    atol: float = 1e-8
    rtol: float = 1e-6
    ls_init_stepsize: float = 0.01
    ls_max_stepsize: float = 1.0
    ls_coef: float = 0.01
    ls_decrease_factor: float = 0.7
    ls_increase_factor: float = 1.3
    ls_reset_option: str = "increase"
    ls_maxls: int = 4
    # extension key (not in the reference schema): "f32" (default), "f16" (SPEC.md §9) or "f32x3" (§9b: layer-2 contractions as three-limb
    # bf16 splits on the matrix pipe, f32-level accuracy); all three are reproduced bit for bit by the oracle
    mlp_dtype: str = "f32"
    math_mode: str = "exact"

    @property
    def num_motors(self) -> int:
        return len(self.input_id)

    @property
    def time_steps(self) -> np.ndarray:
        """cfg_dict['_time_steps'] of the reference (sde_control.py:167,174): dt per MPC step."""
        n_short = min(self.num_short_dt, self.horizon)
        ts = [self.short_step_dt] * n_short + [self.long_step_dt] * (self.horizon - n_short)
        return np.asarray(ts, dtype=np.float32)

    def replace(self, **kw) -> "MPCConfig":
        import dataclasses
        return dataclasses.replace(self, **kw)

    def to_cfg(self):
        """Build the C struct; returns (cfg, keepalive) — keepalive owns the time_steps buffer."""
        m = self.num_motors
        if not (1 <= m <= MAX_MOTORS):
            raise ValueError(f"num_motors {m} out of range")
        if len(self.uref) != m or len(self.input_bound) != m:
            raise ValueError("uref / input_bound length must equal len(input_id)")
        ts = np.ascontiguousarray(self.time_steps)
        c = SdempcCfg()
        c.struct_size = C.sizeof(SdempcCfg)
        c.horizon, c.num_particles, c.num_motors = int(self.horizon), int(self.num_particles), m
        c.time_steps = ts.ctypes.data_as(C.POINTER(C.c_float))
        c.discount = self.discount
        for j in range(m):
            c.uref[j] = self.uref[j]
            lo, hi = self.input_bound[j]
            if not self.enforce_ubound:
                lo, hi = -3.0e38, 3.0e38
            c.u_lo[j], c.u_hi[j] = lo, hi
        c.uerr = self.uerr
        for i in range(3):
            c.perr[i], c.verr[i], c.qerr[i], c.werr[i] = self.perr[i], self.verr[i], self.qerr[i], self.werr[i]
        c.res_mult = self.res_mult
        c.u_slew_coeff = self.u_slew_coeff
        c.has_slew_constr = 0
        if self.u_slew_constr is not None:
            if len(self.u_slew_constr) != m:
                raise ValueError("u_slew_constr must have one [lo, hi] pair per motor")
            c.has_slew_constr = 1
            for j in range(m):
                c.u_slew_lo[j], c.u_slew_hi[j] = self.u_slew_constr[j]
            c.u_slew_constr_coeff = self.u_slew_constr_coeff
        c.max_iter, c.max_no_improvement_iter = int(self.max_iter), int(self.max_no_improvement_iter)
        c.use_moment_scale = 0 if self.moment_scale is None else 1
        c.moment_scale = 0.0 if self.moment_scale is None else float(self.moment_scale)
        c.beta_init, c.atol, c.rtol, c.stepsize = self.beta_init, self.atol, self.rtol, self.stepsize
        c.ls_init_stepsize, c.ls_max_stepsize, c.ls_coef = self.ls_init_stepsize, self.ls_max_stepsize, self.ls_coef
        c.ls_decrease_factor, c.ls_increase_factor = self.ls_decrease_factor, self.ls_increase_factor
        if self.ls_reset_option not in ("increase", "conservative"):
            raise ValueError(f"linesearch.reset_option must be increase|conservative, got {self.ls_reset_option!r}")
        c.ls_reset_option = 1 if self.ls_reset_option == "increase" else 0
        c.ls_maxls = int(self.ls_maxls)
        c.num_state_constr = 0
        if self.state_id:
            ids = [int(i) for i in self.state_id]
            if len(ids) != len(self.state_penalty or []) or len(ids) != len(self.state_bound or []):
                raise ValueError("state_constr: state_id, state_penalty and state_bound must have the same length")
            if any(i < 0 or i > 12 for i in ids) or any(b <= a for a, b in zip(ids, ids[1:])):
                raise ValueError("state_constr.state_id must be strictly ascending indices into the 13-state")
            c.num_state_constr = len(ids)
            for k, i in enumerate(ids):
                c.state_id[k] = i
                c.state_w[k] = float(np.float32(self.state_penalty[k]) * np.float32(self.constr_pen))
                c.state_lo[k], c.state_hi[k] = float(self.state_bound[k][0]), float(self.state_bound[k][1])
        if self.mlp_dtype not in ("f32", "f16", "f32x3"):
            raise ValueError(f"mlp_dtype must be f32|f16|f32x3, got {self.mlp_dtype!r}")
        c.mlp_dtype = {"f32": 0, "f16": 1, "f32x3": 2}[self.mlp_dtype]
        if self.math_mode not in ("exact", "fast"):
            raise ValueError(f"math_mode must be exact|fast, got {self.math_mode!r}")
        c.math_mode = 1 if self.math_mode == "fast" else 0
        return c, ts


def mpc_config_from_dict(d: dict) -> MPCConfig:
    cfg = MPCConfig()
    cfg.learned_model_params = d.get("learned_model_params")
    cfg.trajectory_path = d.get("trajectory_path")
    ic = d.get("input_constr", {})
    if ic:
        cfg.input_id = [int(i) for i in ic["input_id"]]
        cfg.input_bound = [[float(a), float(b)] for a, b in ic["input_bound"]]
    cfg.enforce_ubound = bool(d.get("enforce_ubound", True))
    if d.get("state_constr"):
        sc = d["state_constr"]
        if sc.get("slack_proximal", False):
            raise NotImplementedError("state_constr.slack_proximal: True (slack variables as extra decision variables) is not supported; "
                                      "the penalty form (slack_proximal: False) is (SPEC.md §5.3)")
        cfg.state_id = [int(i) for i in sc["state_id"]]
        cfg.state_penalty = [float(v) for v in sc["state_penalty"]]
        cfg.state_bound = [[float(a), float(b)] for a, b in sc["state_bound"]]
        cfg.constr_pen = float(sc.get("constr_pen", 1.0))     # slack_scaling only concerns the slack-variable form
    cp = d.get("cost_params", {})
    m = len(cfg.input_id)
    uref = np.atleast_1d(np.asarray(cp.get("uref", [0.0] * m), dtype=np.float32))
    cfg.uref = [float(x) for x in (np.repeat(uref, m) if uref.size == 1 else uref)]
    cfg.uerr = float(cp.get("uerr", 0.0))
    cfg.perr, cfg.verr = _vec3(cp.get("perr", 0.0)), _vec3(cp.get("verr", 0.0))
    cfg.qerr, cfg.werr = _vec3(cp.get("qerr", 0.0)), _vec3(cp.get("werr", 0.0))
    cfg.res_mult = float(cp.get("res_mult", 0.0))
    cfg.u_slew_coeff = float(cp.get("u_slew_coeff", 0.0))
    if "u_slew_constr" in cp:
        cfg.u_slew_constr = [[float(a), float(b)] for a, b in cp["u_slew_constr"]]
        cfg.u_slew_constr_coeff = float(cp.get("u_slew_constr_coeff", 0.0))
    for k in ("horizon", "num_short_dt", "num_particles"):
        if k in d:
            setattr(cfg, k, int(d[k]))
    for k in ("short_step_dt", "long_step_dt", "discount"):
        if k in d:
            setattr(cfg, k, float(d[k]))
    apg = d.get("apg_mpc", {})
    for k in ("stepsize", "beta_init", "atol", "rtol"):
        if k in apg:
            setattr(cfg, k, float(apg[k]))
    for k in ("max_iter", "max_no_improvement_iter"):
        if k in apg:
            setattr(cfg, k, int(apg[k]))
    ms = apg.get("moment_scale")
    cfg.moment_scale = None if ms is None else float(ms)
    ls = apg.get("linesearch")
    if ls:
        cfg.ls_init_stepsize = float(ls.get("init_stepsize", cfg.ls_init_stepsize))
        cfg.ls_max_stepsize = float(ls.get("max_stepsize", cfg.ls_max_stepsize))
        cfg.ls_coef = float(ls.get("coef", cfg.ls_coef))
        cfg.ls_decrease_factor = float(ls.get("decrease_factor", cfg.ls_decrease_factor))
        cfg.ls_increase_factor = float(ls.get("increase_factor", cfg.ls_increase_factor))
        cfg.ls_reset_option = str(ls.get("reset_option", cfg.ls_reset_option))
        cfg.ls_maxls = int(ls.get("maxls", cfg.ls_maxls))
    else:
        cfg.ls_maxls = 0
    cfg.mlp_dtype = str(d.get("mlp_dtype", "f32"))
    cfg.math_mode = str(d.get("math_mode", "exact"))
    return cfg


def load_mpc_config(path: str) -> MPCConfig:
    with open(os.path.expanduser(path)) as f:
        return mpc_config_from_dict(yaml.safe_load(f))
