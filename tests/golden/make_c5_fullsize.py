"""Self-generated golden vectors (NOT reference output: the reference's arithmetic is not in /root/reference, SPEC.md): ONE full-length solve of
BASELINE config 5 (Iris H=200 P=1024, 200 iterations, ~400 line-search rollouts) by the CPU oracle in the two matrix-pipe arithmetics —
`mlp_dtype: f16` (the mode BASELINE.json names for this config) and `f32x3` — through the instruction model of SPEC.md §9a, in either math mode
(`fast`: through the model of the transcendental instructions, §10a — the arithmetic bench.py's C5 legs time), and ONE full-length solve of config 3
(Hexa H=50 P=256) in the arithmetic bench.py times it in (`f32x3` / `fast`).
The scalar oracle needs 10 - 60 minutes per C5 solve on one core (its particle loops spread over the cores given: same bits), which is why the GPU suite and
bench.py compare against these committed results instead of recomputing them (tests/test_gpu_parity.py::test_full_length_solves_match_the_committed_oracle_results;
benchlib/legs.py plants the instance into its timed batch).
usage: python tests/golden/make_c5_fullsize.py f16|f32x3 [exact|fast] [c5|c3] [threads]
       writes tests/golden/<config>_fullsize_<mlp>[_fast].npz"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import load_mpc_config, synthetic_hexa, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W

INSTANCE_SEED, KEY_INDEX, CURR_T = 3, 3, 0.35      # the instance tests/tools/fullsize_parity.py solves
YAML = {"c5": "c5_iris_traj_h200_p1024.yaml", "c3": "c3_hexa_traj_h50_p256.yaml"}
COMMITTED = [("c5", "f16", "exact"), ("c5", "f32x3", "exact"), ("c5", "f16", "fast"), ("c5", "f32x3", "fast"), ("c3", "f32x3", "fast")]


def golden_path(mode, math="exact", config="c5"):
    return os.path.join(ROOT, "tests", "golden", f"{config}_fullsize_{mode}{'_fast' if math == 'fast' else ''}.npz")


def model_of(config):
    return synthetic_hexa() if config == "c3" else synthetic_iris()


def problem(mode, math="exact", config="c5"):
    cfg = load_mpc_config(os.path.join(ROOT, "configs", YAML[config])).replace(mlp_dtype=mode, math_mode=math)
    x0 = W.random_initial_states(1, INSTANCE_SEED)
    xref = np.stack([W.reference_window(CURR_T, cfg.time_steps)])
    key = prng.split(prng.PRNGKey(10), KEY_INDEX + 1)[KEY_INDEX:KEY_INDEX + 1]
    return cfg, x0, xref, key


if __name__ == "__main__":
    import orc
    mode = sys.argv[1]
    math = sys.argv[2] if len(sys.argv) > 2 else "exact"
    config = sys.argv[3] if len(sys.argv) > 3 else "c5"
    orc.set_threads(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    cfg, x0, xref, key = problem(mode, math, config)
    model = model_of(config)
    O = orc.Oracle(cfg, model)
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    S = SdeMpcSolver(cfg, model, max_batch=1)          # (create / reset are host-only: no GPU needed for the initial guess)
    yk, info0 = S.reset()
    S.close()
    noise = orc.noise_from_key(key[0], cfg.num_particles, cfg.horizon)
    t = time.time()
    uo, xe, io, _ = O.solve(x0[0], xref[0], noise, yk, float(info0["stepsize"]))
    dt = time.time() - t
    np.savez_compressed(golden_path(mode, math, config), uopt=uo, xevol=xe, info=io, u0=yk, stepsize=np.float32(info0["stepsize"]), oracle_seconds=np.float32(dt))
    print(f"{config} full-length solve, mlp_dtype {mode}, math_mode {math}: oracle {dt:.0f} s, N_it {io[2]:.0f}, N_ls {io[7]:.0f}, cost {io[5]:.6g} -> {io[6]:.6g}")
