"""Self-generated fixture (NOT reference output): the four f32 arithmetics of the library — (f32 | f32x3) x (exact | fast) — and the float64 build of
the same oracle on 64 C1-sized (Iris posctrl YAML, H = 20, P = 32, 100 iterations) and 8 C2-sized (Iris traj YAML, H = 50, P = 128, 200 iterations)
problem instances: ONE gradient at a perturbed control sequence and the FULL cold-start solve, per instance and arithmetic.
tests/test_arithmetic_referee_cpu.py recomputes every gradient and a sample of the solves and asserts the referee's criteria on the whole table;
profiles/r5_referee.json is the table itself. About 75 CPU-minutes (the f32x3 solves of the C2-sized instances take the matrix-instruction model two minutes each):
usage: python tests/golden/make_referee.py [threads]      (writes tests/golden/referee.npz, profiles/r5_referee.json)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from benchlib import referee as R
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W

SETS = {"c1": ("c1_iris_posctrl_h20_p32.yaml", 64, 7000, 11), "c2": ("c2_iris_traj_h50_p128.yaml", 8, 7100, 12)}      # yaml, instances, state seed, key seed


def problems(name):
    """the instances of a set: inputs are regenerated from seeds, only results are stored"""
    yaml, n, seed, kseed = SETS[name]
    cfg = load_mpc_config(os.path.join(ROOT, "configs", yaml))
    H, m = cfg.horizon, cfg.num_motors
    x0 = W.random_initial_states(n, seed)
    if "posctrl" in yaml:
        xref = np.stack([W.constant_reference(W.HOVER, H)] * n)
    else:
        xref = np.stack([W.reference_window(0.05 * ((7 * b) % 160), cfg.time_steps) for b in range(n)])
    keys = prng.split(prng.PRNGKey(kseed), n)
    uref = np.asarray(cfg.uref, np.float32)
    u0 = np.tile(uref[None, None], (n, H, 1)).astype(np.float32)                                   # cold start: the hover guess (what m_reset returns)
    ug = np.clip(u0 + 0.1 * np.random.default_rng(seed).standard_normal((n, H, m)), 1e-4, 1).astype(np.float32)      # where the one gradient is taken
    return cfg, x0, xref, keys, u0, ug


def oracle_for(cfg, model, arith):
    if arith == "f64":
        return orc.Oracle(cfg.replace(mlp_dtype="f32", math_mode="exact"), model, double=True)
    mlp, mm = arith.split("/")
    return orc.Oracle(cfg.replace(mlp_dtype=mlp, math_mode=mm), model)


ARITHS = ["f64"] + [R.name(*a) for a in R.ARITHMETICS]


def compute(name, n_threads, which=None, solves=True, log=print):
    """{arith: (grad f64[n,H,m], grad cost f64[n], uopt f32[n,H,m] or None)} for instances `which` (default all)"""
    cfg, x0, xref, keys, u0, ug = problems(name)
    model = synthetic_iris()
    idx = list(range(len(x0))) if which is None else list(which)
    P, H = cfg.num_particles, cfg.horizon
    noise_of = lambda i: orc.noise_from_key(keys[i], P, H)
    out = {}
    for a in ARITHS:
        t = time.time()
        g = R.run_threads([(lambda i=i: oracle_for(cfg, model, a).grad(x0[i], ug[i], xref[i], noise_of(i))) for i in idx], n_threads)
        u = None
        if solves:
            # the most expensive first: one solve per thread; a tail shorter than the thread count would idle cores, so the C2 set (8 instances) is one round
            u = R.run_threads([(lambda i=i: oracle_for(cfg, model, a).solve(x0[i], xref[i], noise_of(i), u0[i], cfg.ls_init_stepsize)[0]) for i in idx], n_threads)
        out[a] = (np.stack([gi for _, gi in g]), np.array([c for c, _ in g]), np.stack(u) if solves else None)
        log(f"{name} {a}: {len(idx)} instances, {time.time() - t:.0f} s")
    return out


def table_of(res):
    grad_rows = {a: [R.gradient_error(res[a][0][i], res["f64"][0][i], res[a][1][i], res["f64"][1][i]) for i in range(len(res[a][0]))] for a in ARITHS[1:]}
    solve_rows = {a: [R.solve_error(res[a][2][i], res["f64"][2][i]) for i in range(len(res[a][2]))] for a in ARITHS[1:]} if res["f64"][2] is not None else {}
    return R.summarize(grad_rows, solve_rows)


if __name__ == "__main__":
    nthr = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    store, tables = {}, {}
    for name in SETS:
        res = compute(name, nthr)
        for a in ARITHS:
            k = a.replace("/", "_")
            store[f"{name}_{k}_grad"], store[f"{name}_{k}_gcost"], store[f"{name}_{k}_uopt"] = res[a]
        tables[name] = table_of(res)
        tables[name + "_ratios_to_f32_exact"] = R.ratios_to(tables[name])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "referee.npz"), **store)
    tables["note"] = ("oracle results (the GPU reproduces each f32 arithmetic bit for bit) against the float64 build of the same oracle; gradients relative to the "
                      "float64 gradient's largest entry; solves: cold start, all controls within abs + rel 1e-4 of the float64 solve of the same instance")
    json.dump(tables, open(os.path.join(ROOT, "profiles", "r5_referee.json"), "w"), indent=1)
    print(json.dumps(tables, indent=1))
