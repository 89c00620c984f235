"""GPU tests of the Python boundary: the reference-shaped callables and the ROS-free worker loop."""
import os

import numpy as np
import pytest

import orc
from cases import CDIR, bits_differ
from sde4mbrl_px4_amd import jax_shim
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.sde_mpc_design import _next_key, load_mpc_problem
from sde4mbrl_px4_amd.utils import enu2ned
from worker import CONTROL_STATE, KEY2INDEX_INFO, MpcWorker, select_command

pytestmark = pytest.mark.gpu
SMALL = dict(max_iter=6, max_no_improvement_iter=6)


def test_m_mpc_matches_oracle_on_identical_seed():
    prob = load_mpc_problem(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), overrides=SMALL)
    prob.shift_warm_start = False
    x = W.random_initial_states(1, 4)[0]
    xdes = W.HOVER.copy()
    rng = jax_shim.random.PRNGKey(10)                                    # launch seed (iris_sdectrl.launch:8)
    st = prob.m_reset(x=x, rng=rng, xdes=xdes)
    uopt, st2, rng2, xevol = prob.m_mpc(x, rng, st, curr_t=0.0, xdes=xdes)
    uopt.block_until_ready()
    assert uopt.shape == (20, 4) and xevol.shape == (21, 13) and uopt.dtype == np.float32
    new_rng, sub = _next_key(rng)
    assert np.array_equal(np.stack([new_rng, sub]), orc.split(rng, 2)) and np.array_equal(rng2, new_rng)   # threefry split (SPEC.md §7.3)
    noise = orc.noise_from_key(sub, 32, 20)
    O = orc.Oracle(prob.cfg, prob.model)
    # frame contract (SPEC.md §1a): x arrives NED and is flipped into the solver's ENU frame, xdes is ENU, xevol comes back in x's frame
    assert prob.convert_to_enu
    uo, xe, info, _ = O.solve(enu2ned(x, np), W.constant_reference(xdes, 20), noise, np.asarray(st.yk), float(st.stepsize))
    xe = np.stack([enu2ned(r, np) for r in xe])
    np.testing.assert_allclose(uopt, uo, rtol=1e-4, atol=1e-6)
    assert bits_differ(uopt, uo) == 0 and bits_differ(xevol, xe) == 0
    np.testing.assert_allclose(xevol[0], x, atol=2e-7)                    # the predicted trajectory starts at the reported state
    assert float(st2.num_steps) == info[2] and float(st2.opt_cost) == info[6] and float(st2.init_cost) == info[5]
    assert not np.array_equal(rng, rng2)
    # same key -> same solution (identical seeds), new key -> different noise
    uopt_b, _, _, _ = prob.m_mpc(x, rng, st, curr_t=0.0, xdes=xdes)
    assert bits_differ(uopt, uopt_b) == 0


def test_worker_replays_mode_sequence():
    traj = load_mpc_problem(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"), horizon=12, num_particles=32, trajectory=W.lemniscate_state, overrides=SMALL)
    pos = load_mpc_problem(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), horizon=10, overrides=SMALL)
    wk = MpcWorker(traj, pos, seed=10)
    wk.warm_up()
    cs = CONTROL_STATE
    x = W.random_initial_states(1, 1)[0]
    tgt = W.lemniscate_state(0.0)
    t_us = 1_000_000.0
    seq = [("none", 0.0), ("pos", 0.0), ("idle", 0.0), ("idle", 0.0), ("idle", 0.0), ("traj", 0.0), ("traj", 0.05)]
    traj_states = []
    for mode, dur in seq:
        uopt, wopt, st = wk.step(x, cs[mode], dur, tgt, t_us)
        H = 12 if mode == "traj" else 10
        assert uopt.shape == (H, 4) and wopt.shape == (H, 4)
        np.testing.assert_allclose(wopt[:, 0], uopt.mean(axis=1), rtol=1e-6)
        assert uopt.min() >= 1e-4 and uopt.max() <= 1.0
        sh = wk.shared
        np.testing.assert_array_equal(sh.u_opt[:H], uopt)
        np.testing.assert_array_equal(sh.w_opt[:H], wopt.astype(np.float64))      # f32 values stored in the f64 block (:633-635)
        assert sh.w_opt.dtype == np.float64
        assert sh.opt_info[KEY2INDEX_INFO["sample_time_posmpc"]] == np.float32(t_us)
        assert sh.opt_info[KEY2INDEX_INFO["num_steps"]] == float(st.num_steps)
        if mode != "idle":                       # in idle the logged state is the traj solver's (:435), solved every other tick
            assert float(st.num_steps) > 0
        assert sh.opt_info[KEY2INDEX_INFO["costT"]] <= sh.opt_info[KEY2INDEX_INFO["cost0"]]
        traj_states.append(float(wk.opt_state_traj.opt_cost))
        t_us += 50_000.0
    # idle alternation (sde_control.py:406-408): entering idle resets the traj solver and sets the toggle, so the
    # traj problem is solved on the 2nd idle tick, skipped on the 3rd
    assert traj_states[2] == 0.0 and traj_states[3] > 0.0 and traj_states[4] == traj_states[3]
    sel = select_command(t_us + 60_000.0, t_us, wk.dt_usec_traj, wk.shared.u_opt, wk.shared.w_opt, 12)
    assert sel[0] == 1 and sel[1].shape == (6,)


FORK_SCRIPT = r"""
import os, sys, json, multiprocessing as mp
sys.path.insert(0, os.environ["REPO"])
os.environ["SDEMPC_PREFORK"] = "shape"
import numpy as np
from sde4mbrl_px4_amd import jax_shim as jax
from sde4mbrl_px4_amd.sde_mpc_design import load_mpc_from_cfgfile
from sde4mbrl_px4_amd.workload import HOVER, random_initial_states

cfg_dict, (m_reset, m_mpc), _, _ = load_mpc_from_cfgfile(os.path.join(os.environ["REPO"], "configs", "c1_iris_posctrl_h20_p32.yaml"),
                                                       overrides=dict(max_iter=5, max_no_improvement_iter=5))
x0 = HOVER.copy(); rng = jax.random.PRNGKey(10)
reset_c = jax.jit(m_reset).lower(x=x0, rng=rng, xdes=x0).compile()           # parent, pre-fork (sde_control.py:702-707)
st = reset_c(x=x0, rng=rng, xdes=x0)
mpc_c = jax.jit(m_mpc).lower(x0, rng, st, curr_t=0.01, xdes=x0).compile()     # :713
u_probe, _, _, _ = mpc_c(x0, rng, st, curr_t=0.01, xdes=x0)                    # :717 warm-up -> shape probe, no HIP

def worker(q):                                                                 # mpc_process_fn (:328) in the forked child
    x = random_initial_states(1, 2)[0]
    u, st2, r2, xe = mpc_c(x, rng, st, curr_t=0.0, xdes=HOVER)
    u.block_until_ready()
    q.put({"shape": list(np.array(u).shape), "num_steps": float(st2.num_steps), "opt": float(st2.opt_cost), "init": float(st2.init_cost),
           "moved": bool(np.abs(np.array(u) - 0.71).max() > 1e-6)})

ctx = mp.get_context("fork")
q = ctx.Queue(); p = ctx.Process(target=worker, args=(q,)); p.start(); res = q.get(timeout=120); p.join(30)
print(json.dumps({"probe_is_warm_start": bool(np.all(np.array(u_probe) == np.float32(0.71))), "child": res, "exit": p.exitcode}))
"""


def test_forked_worker_solves_after_prefork_shape_probe(tmp_path):
    """The reference's process model: solvers built and warmed up in the parent, used in the forked mpc_process."""
    import json
    import subprocess
    import sys
    from cases import ROOT
    f = tmp_path / "fork_flow.py"
    f.write_text(FORK_SCRIPT)
    out = subprocess.run([sys.executable, str(f)], env=dict(os.environ, REPO=ROOT), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["probe_is_warm_start"] and res["exit"] == 0
    assert res["child"]["shape"] == [20, 4] and res["child"]["num_steps"] == 5 and res["child"]["moved"] and res["child"]["opt"] <= res["child"]["init"]


def test_bench_two_ranks_self_started_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts both ranks before anything touches the GPU, rank 0's single JSON
    line comes back, n_gpus == 2. Both ranks share GPU 0 here (SDEMPC_BENCH_DEVICE) and rendezvous over gloo (SDEMPC_BENCH_BACKEND):
    the one-GPU box has neither a second GPU nor an RCCL peer; the driver's multi-GPU runs use one GPU per rank over RCCL / xGMI."""
    import json
    import subprocess
    import sys
    from cases import ROOT
    env = dict(os.environ, SDEMPC_BENCH_DEVICE="0", SDEMPC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64", "--max-iter", "12",
                          "--latency-reps", "2", "--latency-warmup", "1", "--no-cpu-baseline", "--verify", "2", "--c4-reps", "6"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["instances_per_gpu"] == 64
    assert rec["verified_instances"] == 2 and rec["verified_bit_exact"] is True
    assert rec["roofline"]["frac"] > 0 and rec["roofline"]["kernel_ms"] > 0
    assert rec["cpu_baseline"].startswith("skipped")                      # stated, not silently absent, in N > 1 lines
    assert "over_gpus" in rec["power"]                                    # rank 0 samples every GPU of the job; the spread is reported (None: region too short for a sample)
    c4 = rec["c4_one_instance_per_gpu"]                                   # BASELINE config 4: one instance per rank, barrier-aligned ticks
    # (the two ranks share ONE GPU here: their cooperative grids cannot be co-resident, a tick whose barrier gave up is dropped, not timed)
    assert c4["instances"] == 2 and c4["ticks"] + c4["ticks_dropped"] == 6 and c4["ticks"] >= 1 and c4["p50_tick_ms"] > 0 and c4["value"] > 0
    # EVERY rank proved its own results: two instances of its own timed launch and the instance of its last good config-4 tick, against the oracle
    vr = rec["verified_by_rank"]
    assert [v["rank"] for v in vr] == [0, 1] and all(v["asked"] in (2, 3) and v["checked_bit_exact"] == v["asked"] for v in vr), vr
    assert c4.get("verified_bit_exact") in (True, None)                   # (None: no tick of rank 0 completed without a give-up on the shared GPU)


def test_bench_rccl_branch_on_one_gpu():
    """The RCCL code path of bench.py / dist.py on the one GPU a test box has: SDEMPC_BENCH_FORCE_DIST=1 makes a single rank initialise the
    process group with backend nccl (= RCCL on ROCm) and run the device-tensor broadcast of the model blob, the all-reduce behind
    max_over_ranks, the barriers and destroy_process_group. It proves that the library loads and that the collectives run on device tensors —
    not scaling, which needs the driver's 8-GPU node."""
    import json
    import subprocess
    import sys
    from cases import ROOT
    env = dict(os.environ, SDEMPC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("SDEMPC_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "64", "--max-iter", "12",
                          "--latency-reps", "2", "--latency-warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-tolerance-modes", "--verify", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["verified_bit_exact"] is True
    assert rec["verified_by_rank"] == [{"rank": 0, "checked_bit_exact": 3, "asked": 3}]      # two instances of the timed launch + the config-4 tick (counts gathered through RCCL)
    assert rec["c4_one_instance_per_gpu"]["verified_bit_exact"] is True
    assert rec["c4_one_instance_per_gpu"]["instances"] == 1 and rec["c4_one_instance_per_gpu"]["barrier_give_ups_rank0"] == 0      # (max over ranks through RCCL)
    # the process really loaded RCCL and ran collectives on it
    chk = subprocess.run([sys.executable, "-c", "import torch, torch.distributed as d, os; d.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0)); "
                          "t = torch.ones(4, device='cuda'); d.all_reduce(t); d.broadcast(t, src=0); d.barrier(); print(d.get_backend(), float(t.sum())); d.destroy_process_group()"],
                         env=env, capture_output=True, text=True, timeout=300)
    last = chk.stdout.strip().splitlines()[-1].split()                     # (RCCL prints its own banner lines before ours)
    assert chk.returncode == 0 and last[0] == "nccl" and float(last[1]) == 4.0, (chk.stdout[-500:], chk.stderr[-2000:])


def test_hold_mode_tracks_the_current_state_in_the_right_frame():
    """Mode 'none' of the worker loop: xdes = enu2ned(curr_state) (sde_control.py:400) with the state reported in NED. The solver's
    reference then equals its own initial state: the initial cost is the pure control / uncertainty cost of hovering there, far below the
    cost of the same call with the target left in the wrong frame (what a solver that compared x and xdes frame-blind would track)."""
    prob = load_mpc_problem(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), overrides=SMALL)
    x = W.random_initial_states(1, 9)[0]
    x[0:3] = [1.5, -0.7, -2.0]                                           # NED: 2 m above the ground
    rng = jax_shim.random.PRNGKey(3)
    st = prob.m_reset(x=x, rng=rng, xdes=x)
    _, st_hold, _, xevol = prob.m_mpc(x, rng, st, curr_t=0.0, xdes=enu2ned(x, np))
    _, st_wrong, _, _ = prob.m_mpc(x, rng, st, curr_t=0.0, xdes=x)
    assert float(st_hold.init_cost) < 0.05 * float(st_wrong.init_cost)
    np.testing.assert_allclose(xevol[0], x, atol=2e-7)
    assert abs(float(xevol[1, 2]) - float(x[2])) < 0.05                  # stays near its altitude over the first step (NED z)


def test_worker_replay_matches_the_oracle_driven_fixture():
    """the scripted 34-tick mode sequence (none -> pos -> idle -> traj -> pos -> none -> idle -> traj) through the HIP
    path, compared per tick with the committed fixture an oracle-backed worker produced (tests/golden/make_worker_replay.py): u_opt (f32),
    w_opt (f64), opt_info[1:8] and the rows `select_command` picks at four lags — bit for bit."""
    import replay
    fx = dict(np.load(replay.FIXTURE))
    got = replay.run(oracle=False)
    assert np.array_equal(got["states"], fx["states"]) and np.array_equal(got["modes"], fx["modes"])
    for k in range(len(replay.MODES)):
        assert bits_differ(got["u_opt"][k], fx["u_opt"][k]) == 0, (k, replay.MODES[k])
        assert np.array_equal(got["w_opt"][k].view(np.uint64), fx["w_opt"][k].view(np.uint64)), (k, replay.MODES[k])
        assert bits_differ(got["opt_info"][k], fx["opt_info"][k]) == 0, (k, replay.MODES[k], got["opt_info"][k], fx["opt_info"][k])
    assert np.array_equal(got["sel_idx"], fx["sel_idx"]) and np.array_equal(got["sel_u"], fx["sel_u"]) and np.array_equal(got["sel_w"], fx["sel_w"])


def test_in_process_worker_solves_with_prefork_variable_set(monkeypatch):
    """SDEMPC_PREFORK=shape in the environment of a process that never forks (notebook, in-process MpcWorker): only the first call of a
    jax_shim-compiled callable is a shape probe; the worker and every later call solve for real."""
    monkeypatch.setenv("SDEMPC_PREFORK", "shape")
    traj = load_mpc_problem(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"), horizon=12, num_particles=32, trajectory=W.lemniscate_state, overrides=SMALL)
    pos = load_mpc_problem(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"), horizon=10, overrides=SMALL)
    wk = MpcWorker(traj, pos, seed=10)
    x = W.random_initial_states(1, 5)[0]
    uopt, wopt, st = wk.step(x, CONTROL_STATE["pos"], 0.0, W.HOVER, 1e6)
    assert float(st.num_steps) > 0 and np.abs(uopt - 0.71).max() > 1e-6
    x0 = W.HOVER.copy(); rng = jax_shim.random.PRNGKey(1)
    st0 = pos.m_reset(x=x0, rng=rng, xdes=x0)
    mpc_c = jax_shim.jit(pos.m_mpc).lower(x0, rng, st0, curr_t=0.0, xdes=x0).compile()
    with pytest.warns(RuntimeWarning, match="SHAPE PROBE"):
        u1, s1, _, _ = mpc_c(x, rng, st0, curr_t=0.0, xdes=W.HOVER)      # first call: probe (warm start returned, no iterations, marked as unsolved)
    u2, s2, _, _ = mpc_c(x, rng, st0, curr_t=0.0, xdes=W.HOVER)          # second call: a real solve
    assert float(s1.num_steps) == 0 and np.isnan(float(s1.opt_cost)) and np.all(np.array(u1) == np.float32(0.71))
    assert float(s2.num_steps) > 0 and np.abs(np.array(u2) - 0.71).max() > 1e-6
