"""Importer for pickled parameter trees: turns a nested dict of arrays — the shape in which JAX/Haiku
projects usually pickle their weights — into this build's `RotorSDEModel`.

What the reference fixes: only that `learned_model_params` names a pickle (`launch/iris_sitl_traj_mpc.yaml:3`,
`iris_sitl_sde.pkl`) that the external package sde4mbrl loads. The pickle's layout — module names, array orientation, where
the physics prior lives — is NOT in the reference, so nothing here is guessed: the caller states which leaf feeds which slot
of SPEC.md §2 in a mapping (a dict, or a YAML file beside the pickle), and the importer checks every shape. The architecture
must be the one of SPEC.md §2 (drift MLP (6+m) -> 32 -> 32 -> 6, density MLP 6 -> 32 -> 1); a pickle of a different
architecture is rejected with the offending shape in the message.

Security: a pickle can execute code when loaded. `load_param_tree` uses an unpickler that resolves only numpy array
reconstruction and plain containers; anything else (including jaxlib device arrays, which need JAX to load) raises. Convert
such files once, where JAX is installed, with `jax.device_get` + `pickle.dump`.

Mapping format (all keys optional except the ones without default; paths are '/'-joined keys of the tree):
    drift_l1:   {w: "drift/linear/w",   b: "drift/linear/b",   layout: "in_out"}   # [6+m, 32]; inputs ordered (v_body, omega, u)
    drift_l2:   {w: "drift/linear_1/w", b: "drift/linear_1/b", layout: "in_out"}   # [32, 32]
    drift_out:  {w: "drift/linear_2/w", b: "drift/linear_2/b", layout: "in_out"}   # [32, 6] -> (F_res xyz, tau_res xyz)
    density_l1: {w: "density/linear/w", b: "density/linear/b", layout: "in_out"}   # [6, 32]
    density_out:{w: "density/linear_1/w", b: "density/linear_1/b", layout: "in_out"} # [32, 1]
    physics:    {mass: 1.5, grav: 9.81, inertia: [..3], thrust_poly: [ct2, ct1, ct0], moment_poly: [cm2, cm1],
                 rotor_x: [..m], rotor_y: [..m], rotor_dir: [..m], res_force_scale: [..3], res_torque_scale: [..3], sigma: [..6]}
A physics entry may be a number / list, or a string path into the tree. `layout` is "in_out" (Haiku/Flax: y = x @ w) or
"out_in" (this build, torch: y = w @ x).
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict, Mapping

import numpy as np
import yaml

from .model import RotorSDEModel

_SAFE = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
    ("builtins", "frozenset"), ("builtins", "slice"), ("builtins", "complex"), ("builtins", "bytearray"),
}


class _ArrayTreeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            f"refusing to load {module}.{name}: only numpy arrays inside dict/list/tuple containers are accepted "
            "(convert device arrays with jax.device_get before pickling)")


def load_param_tree(path_or_bytes) -> Any:
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    return _ArrayTreeUnpickler(io.BytesIO(data)).load()


def flatten_tree(tree, prefix="") -> Dict[str, np.ndarray]:
    """{'a/b/w': ndarray, ...}; list / tuple members are addressed by index."""
    out: Dict[str, np.ndarray] = {}
    if isinstance(tree, Mapping):
        for k, v in tree.items():
            out.update(flatten_tree(v, f"{prefix}{k}/"))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            out.update(flatten_tree(v, f"{prefix}{i}/"))
    else:
        out[prefix[:-1]] = np.asarray(tree)
    return out


def _leaf(flat, path, what):
    if path not in flat:
        close = ", ".join(sorted(flat)[:12])
        raise KeyError(f"{what}: no leaf {path!r} in the parameter tree (first leaves: {close})")
    return np.asarray(flat[path], dtype=np.float32)


def _linear(flat, spec, n_in, n_out, what):
    w = _leaf(flat, spec["w"], what + ".w")
    layout = spec.get("layout", "in_out")
    if layout not in ("in_out", "out_in"):
        raise ValueError(f"{what}: layout must be in_out|out_in, got {layout!r}")
    if layout == "in_out":
        w = w.T
    if w.shape != (n_out, n_in):
        raise ValueError(f"{what}: weight has shape {tuple(w.shape)} (as [out, in]), SPEC.md §2 needs {(n_out, n_in)}")
    b = _leaf(flat, spec["b"], what + ".b").reshape(-1) if spec.get("b") else np.zeros(n_out, np.float32)
    if b.shape != (n_out,):
        raise ValueError(f"{what}: bias has shape {tuple(b.shape)}, expected {(n_out,)}")
    return np.ascontiguousarray(w), b


def model_from_param_tree(tree, mapping: Mapping) -> RotorSDEModel:
    flat = flatten_tree(tree)
    phys = dict(mapping.get("physics") or {})

    def pv(name, shape=None):
        if name not in phys:
            raise KeyError(f"mapping.physics lacks {name!r}")
        v = phys[name]
        a = _leaf(flat, v, "physics." + name) if isinstance(v, str) else np.asarray(v, dtype=np.float32)
        if shape is not None and a.reshape(-1).shape != shape:
            raise ValueError(f"physics.{name} has {a.size} entries, expected {shape[0]}")
        return a.reshape(-1) if shape is not None else float(a)

    if "rotor_x" not in phys:
        raise KeyError("mapping.physics lacks 'rotor_x'")
    rx = phys["rotor_x"]
    m = int((_leaf(flat, rx, "physics.rotor_x") if isinstance(rx, str) else np.asarray(rx)).size)
    if not 1 <= m <= 8:
        raise ValueError(f"{m} rotors: this build supports 1..8")
    W1, b1d = _linear(flat, mapping["drift_l1"], 6 + m, 32, "drift_l1")
    W2, b2 = _linear(flat, mapping["drift_l2"], 32, 32, "drift_l2")
    W3, b3 = _linear(flat, mapping["drift_out"], 32, 6, "drift_out")
    W1n, b1n = _linear(flat, mapping["density_l1"], 6, 32, "density_l1")
    w3n, b3n = _linear(flat, mapping["density_out"], 32, 1, "density_out")
    return RotorSDEModel(
        num_motors=m, mass=pv("mass"), grav=pv("grav") if "grav" in phys else 9.81,
        inertia=pv("inertia", (3,)), thrust_poly=pv("thrust_poly", (3,)), moment_poly=pv("moment_poly", (2,)),
        rotor_x=pv("rotor_x", (m,)), rotor_y=pv("rotor_y", (m,)), rotor_dir=pv("rotor_dir", (m,)),
        res_force_scale=pv("res_force_scale", (3,)), res_torque_scale=pv("res_torque_scale", (3,)), sigma=pv("sigma", (6,)),
        W1z=np.ascontiguousarray(np.concatenate([W1[:, :6], W1n], axis=0)), b1=np.concatenate([b1d, b1n]),
        W1u=np.ascontiguousarray(W1[:, 6:]), W2=W2, b2=b2, W3=W3, b3=b3, w3n=w3n[0].copy(), b3n=float(b3n[0]))


def import_sde_pickle(pickle_path: str, mapping) -> RotorSDEModel:
    """mapping: dict, or path of a YAML file with the layout described in this module's docstring."""
    if isinstance(mapping, str):
        with open(mapping) as f:
            mapping = yaml.safe_load(f)
    return model_from_param_tree(load_param_tree(pickle_path), mapping)
