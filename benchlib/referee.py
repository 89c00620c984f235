"""Float64 referee of the f32 arithmetics (f32 | f32x3) x (exact | fast): how far is each from the real-number solution?

The four arithmetics of the library are each compared with their CPU oracle bit for bit — that says the GPU computes what SPEC.md says, not that
one SPEC arithmetic is as good as another. The reference's own path is a fifth rounding (f32 JAX on CPU, sde_control.py:6), so the only defensible
claim for any of them is "no further from the float64 evaluation of the same algorithm than a plain f32 evaluation is". This module measures that:

* per GRADIENT (one adjoint evaluation at a given control sequence): rms and max error against the float64 oracle's gradient (oracle/ built
  -DORC_DOUBLE: the same statements on double with libm activations), relative to the gradient's largest entry; relative error of the cost;
* per FULL SOLVE (max_iter APG iterations with their line-search decisions): the fraction of instances whose controls all lie within
  abs + rel 1e-4 (the north star's tolerance) of the float64 solve of the same instance, median / worst max|du|.

Test infrastructure like the rest of the oracle: used by tests/test_arithmetic_referee_cpu.py (oracle results), by bench.py's `vs_float64`
field and by tools/mode_drift.py (GPU results against the float64 oracle). Nothing under sde4mbrl_px4_amd/ imports it.
"""
import threading

import numpy as np

ARITHMETICS = [("f32", "exact"), ("f32x3", "exact"), ("f32", "fast"), ("f32x3", "fast")]
TOL = 1e-4          # north star: controls within 1e-4 (applied as abs + rel)


def name(mlp, mm):
    return f"{mlp}/{mm}"


def gradient_error(g, g64, c=None, c64=None):
    """errors of one gradient against the float64 one, relative to the float64 gradient's largest entry"""
    g, g64 = np.asarray(g, np.float64), np.asarray(g64, np.float64)
    scale = float(np.abs(g64).max())
    e = (g - g64) / scale
    out = {"rms_rel": float(np.sqrt(np.mean(e * e))), "max_rel": float(np.abs(e).max())}
    if c is not None:
        out["cost_rel"] = abs(float(c) - float(c64)) / abs(float(c64))
    return out


def solve_error(u, u64):
    """controls of one full solve against the float64 solve of the same instance"""
    u, u64 = np.asarray(u, np.float64), np.asarray(u64, np.float64)
    d = np.abs(u - u64)
    return {"within": bool(np.all(d <= TOL + TOL * np.abs(u64))), "max_abs_du": float(d.max())}


def summarize(grad_rows, solve_rows):
    """grad_rows / solve_rows: {arithmetic name: [per-instance dicts]} -> one table row per arithmetic"""
    table = {}
    for k in grad_rows:
        g = grad_rows[k]
        rms = np.array([r["rms_rel"] for r in g])
        row = {"gradients": len(g),
               "grad_rms_rel": float(np.sqrt(np.mean(rms * rms))),          # rms over every entry of every instance (equal sizes)
               "grad_rms_rel_worst_instance": float(rms.max()),
               "grad_max_rel_median": float(np.median([r["max_rel"] for r in g])),
               "grad_max_rel_worst": float(max(r["max_rel"] for r in g))}
        if g and "cost_rel" in g[0]:
            row["cost_rel_median"] = float(np.median([r["cost_rel"] for r in g]))
            row["cost_rel_worst"] = float(max(r["cost_rel"] for r in g))
        s = solve_rows.get(k) or []
        if s:
            du = np.array([r["max_abs_du"] for r in s])
            row.update({"solves": len(s), "solves_within_1e-4_of_float64": float(np.mean([r["within"] for r in s])),
                        "max_abs_du_median": float(np.median(du)), "max_abs_du_worst": float(du.max())})
        table[k] = row
    return table


def ratios_to(table, base="f32/exact", keys=("grad_rms_rel", "grad_max_rel_worst")):
    """every arithmetic's per-gradient figures as multiples of the base arithmetic's (the assertion of the CPU test: <= 1.5)"""
    return {k: {q: (row[q] / table[base][q] if table[base][q] > 0 else float("inf")) for q in keys} for k, row in table.items() if k != base}


def run_threads(jobs, n_threads):
    """jobs: list of zero-argument callables (ctypes calls release the GIL); returns their results in order"""
    out = [None] * len(jobs)
    errs = []
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                j = nxt[0]; nxt[0] += 1
            if j >= len(jobs):
                return
            try:
                out[j] = jobs[j]()
            except Exception as e:          # noqa: BLE001
                with lock:
                    errs.append(f"job {j}: {type(e).__name__}: {e}")
    th = [threading.Thread(target=work) for _ in range(max(1, min(n_threads, len(jobs))))]
    [t.start() for t in th]; [t.join() for t in th]
    if errs:
        raise RuntimeError("; ".join(errs[:3]))
    return out


class Float64Referee:
    """float64 gradient / full solve of given instances by the oracle's -DORC_DOUBLE build (threads; one oracle object per job)"""

    def __init__(self, orc, cfg, blob):
        self.orc, self.blob = orc, blob
        self.cfg = cfg.replace(mlp_dtype="f32", math_mode="exact")      # the float64 build has ONE arithmetic: double, libm activations, no operand quantisation

    def gradients(self, x0, u, xref, noise_of, n_threads):
        def job(i):
            return lambda: self.orc.Oracle(self.cfg, self.blob, double=True).grad(x0[i], u[i], xref[i], noise_of(i))
        return run_threads([job(i) for i in range(len(x0))], n_threads)

    def solves(self, x0, xref, noise_of, u0, s0, n_threads):
        def job(i):
            return lambda: self.orc.Oracle(self.cfg, self.blob, double=True).solve(x0[i], xref[i], noise_of(i), u0[i], s0)[:3]
        return run_threads([job(i) for i in range(len(x0))], n_threads)
