"""Self-generated golden vectors (NOT reference output: the reference's arithmetic is not in /root/reference, SPEC.md): ONE full-length solve of
BASELINE config 5 (Iris H=200 P=1024, 200 iterations, ~400 line-search rollouts) by the CPU oracle in the two matrix-pipe arithmetics —
`mlp_dtype: f16` (the mode BASELINE.json names for this config) and `f32x3` — through the instruction model of SPEC.md §9a.
The scalar oracle needs 10 - 30 minutes per solve on one core, which is why the GPU suite compares against this committed result instead of
recomputing it (tests/test_gpu_parity.py::test_c5_full_length_solve_matches_the_committed_oracle_result).
usage: python tests/golden/make_c5_fullsize.py f16 | f32x3      (writes tests/golden/c5_fullsize_<mode>.npz)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W

INSTANCE_SEED, KEY_INDEX, CURR_T = 3, 3, 0.35      # the instance tests/tools/fullsize_parity.py solves


def problem(mode):
    cfg = load_mpc_config(os.path.join(ROOT, "configs", "c5_iris_traj_h200_p1024.yaml")).replace(mlp_dtype=mode)
    x0 = W.random_initial_states(1, INSTANCE_SEED)
    xref = np.stack([W.reference_window(CURR_T, cfg.time_steps)])
    key = prng.split(prng.PRNGKey(10), KEY_INDEX + 1)[KEY_INDEX:KEY_INDEX + 1]
    return cfg, x0, xref, key


if __name__ == "__main__":
    mode = sys.argv[1]
    cfg, x0, xref, key = problem(mode)
    model = synthetic_iris()
    O = orc.Oracle(cfg, model)
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    S = SdeMpcSolver(cfg, model, max_batch=1)          # (create / reset are host-only: no GPU needed for the initial guess)
    yk, info0 = S.reset()
    S.close()
    noise = orc.noise_from_key(key[0], cfg.num_particles, cfg.horizon)
    t = time.time()
    uo, xe, io, _ = O.solve(x0[0], xref[0], noise, yk, float(info0["stepsize"]))
    dt = time.time() - t
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"c5_fullsize_{mode}.npz"), uopt=uo, xevol=xe, info=io, u0=yk, stepsize=np.float32(info0["stepsize"]),
                        oracle_seconds=np.float32(dt))
    print(f"c5 full-length solve, mlp_dtype {mode}: oracle {dt:.0f} s, N_it {io[2]:.0f}, N_ls {io[7]:.0f}, cost {io[5]:.6g} -> {io[6]:.6g}")
