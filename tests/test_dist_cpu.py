"""world_size-2 gloo test of the multi-GPU plumbing (weights broadcast, instance sharding, gather)."""
import os
import socket
import subprocess
import sys

import numpy as np

from cases import ROOT
from sde4mbrl_px4_amd.dist import shard_range

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import numpy as np, torch, torch.distributed as dist
from sde4mbrl_px4_amd import MPCConfig, synthetic_iris
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.dist import broadcast_blob, shard_range, max_over_ranks, gather_rows
import orc
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
blob = synthetic_iris().to_blob() if rank == 0 else b""
blob = broadcast_blob(blob, src=0)
cfg = MPCConfig(horizon=6, num_short_dt=6, num_particles=8, u_slew_coeff=1.0)
total = 7
lo, hi = shard_range(total, rank, world)
x0 = W.random_initial_states(total, 0)[lo:hi]
noise = np.concatenate([W.make_noise(1, 8, 6, b) for b in range(lo, hi)])
O = orc.Oracle(cfg, blob)          # CPU stand-in for the per-rank solver (checker only)
u = np.full((6, 4), 0.71, np.float32)
costs = np.array([[O.rollout(x0[i], u, W.reference_window(0.0, cfg.time_steps), noise[i])[0]] for i in range(hi - lo)], np.float32)
allc = gather_rows(costs)
tmax = max_over_ranks(1.0 + rank)
from sde4mbrl_px4_amd.dist import max_over_ranks_each
ticks = max_over_ranks_each([1.0 + rank, 5.0 - 3 * rank, 2.0])       # per-tick durations of barrier-aligned ticks: the slowest rank's each
if rank == 0:
    print(json.dumps({"n": int(allc.shape[0]), "costs": [float(c) for c in allc[:, 0]], "tmax": tmax, "ticks": ticks, "bloblen": len(blob)}))
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_broadcast_shard_gather(tmp_path):
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REPO=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["n"] == 7 and res["tmax"] == 2.0 and res["ticks"] == [2.0, 5.0, 2.0] and res["bloblen"] == 4 * (16 + 2120)
    # single-process reference: same instances, same order
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from sde4mbrl_px4_amd import MPCConfig, synthetic_iris
    from sde4mbrl_px4_amd import workload as W
    cfg = MPCConfig(horizon=6, num_short_dt=6, num_particles=8, u_slew_coeff=1.0)
    O = orc.Oracle(cfg, synthetic_iris())
    u = np.full((6, 4), 0.71, np.float32)
    x0 = W.random_initial_states(7, 0)
    ref = [O.rollout(x0[b], u, W.reference_window(0.0, cfg.time_steps), W.make_noise(1, 8, 6, b)[0])[0] for b in range(7)]
    np.testing.assert_array_equal(np.array(res["costs"], np.float32), np.array(ref, np.float32))


def test_bench_self_start_stops_all_ranks_when_one_fails():
    """`bench.py --gpus 2` without a launcher starts its own ranks before touching the GPU; a rank that cannot run (here: no GPU in the
    container, every rank exits with "needs a GPU") must end the whole job promptly with a non-zero code instead of leaving the others in
    the rendezvous, and nothing may reach stdout (the driver parses ONE JSON line from it)."""
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU: with one the two ranks would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs a GPU" in r.stderr
    assert time.time() - t0 < 90
    # the parent's own summary line names a failing rank and quotes what that rank said
    assert "bench.py: rank " in r.stderr and "exited with code" in r.stderr and "Last lines of that rank's stderr" in r.stderr


def _bench_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **kw)
    return env


def test_bench_four_ranks_dry_run_over_gloo():
    """The rank plumbing of `bench.py --gpus 4` on the CPU (SDEMPC_BENCH_DRY=1: self-start, rendezvous on 127.0.0.1, blob broadcast, all-reduce MAX
    incl. a dropped tick, barriers, destroy) — what the first real multi-GPU run exercises before any kernel is launched. One JSON line, exit 0."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_bench_env(SDEMPC_BENCH_DRY="1"), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 4, "blob_bytes": 4 * (16 + 2120), "max_over_ranks": 4.0, "ticks": [3.0, None],
                   "same_on_every_rank": ["library build", "model blob"], "verified_by_rank": [[2, 2]] * 4, "bad_words_all_ranks": 0}


def test_bench_four_ranks_a_rank_with_a_corrupted_blob_stops_the_run_and_is_named():
    """Rank 2's copy of the broadcast model blob has one bit flipped (SDEMPC_BENCH_CORRUPT_RANK): the fingerprints every rank exchanges at start-up
    disagree, every rank exits non-zero before a single launch, nothing is printed on stdout and the message names rank 2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_bench_env(SDEMPC_BENCH_DRY="1", SDEMPC_BENCH_CORRUPT_RANK="2"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "model blob: rank(s) [2] hold another one than the other ranks" in r.stderr, r.stderr[-1500:]


def test_bench_four_ranks_a_rank_whose_outputs_differ_from_the_oracle_fails_the_run():
    """Every rank checks instances of ITS OWN launches; one flipped bit in rank 3's outputs (SDEMPC_BENCH_CORRUPT_OUTPUT_RANK) — what a wrong device
    binding or a faulty GPU would look like, invisible to rank 0's sample — makes the whole job exit non-zero, with rank 3 short in verified_by_rank."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_bench_env(SDEMPC_BENCH_DRY="1", SDEMPC_BENCH_CORRUPT_OUTPUT_RANK="3"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["verified_by_rank"] == [[2, 2], [2, 2], [2, 2], [0, 2]] and out["bad_words_all_ranks"] == 1
    assert "the outputs of rank(s) [3] differ from the oracle" in r.stderr


def test_bench_four_ranks_one_fails_before_the_rendezvous():
    """Rank 2 dies at start-up (what a rank whose GPU cannot be opened does) while the other three wait in the rendezvous: the parent stops them
    within seconds, exits non-zero, prints nothing on stdout and says in ITS OWN last line which rank failed and why."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_bench_env(SDEMPC_BENCH_DRY="1", SDEMPC_BENCH_FAIL_RANK="2"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert time.time() - t0 < 90
    last = [l for l in r.stderr.splitlines() if l.startswith("bench.py: rank ")][-1]
    assert "rank 2 exited with code" in last and "simulated start-up failure" in last


def test_rank_agreement_and_referee_helpers():
    """benchlib/ranks.py: which ranks hold something else than the others (majority; ties: rank 0's value counts); benchlib/referee.py: error measures and the table"""
    from benchlib import referee as R
    from benchlib.ranks import fingerprint, ranks_that_disagree
    assert ranks_that_disagree([[1, 2], [1, 2], [1, 3], [1, 2]]) == {1: [2]}
    assert ranks_that_disagree([[5], [1]]) == {0: [1]} and ranks_that_disagree([[7, 7]] * 3) == {}
    assert ranks_that_disagree([[9], [4], [4], [4]]) == {0: [0]}                       # rank 0 itself can be the odd one
    assert fingerprint(b"abc") == fingerprint("abc") != fingerprint(b"abd") and -2**63 <= fingerprint(b"x") < 2**63
    g64 = np.array([[1.0, -4.0], [2.0, 0.5]])
    e = R.gradient_error(g64 + np.array([[4e-7, 0.0], [0.0, -4e-7]]), g64, 10.0 + 1e-6, 10.0)
    assert abs(e["max_rel"] - 1e-7) < 1e-12 and abs(e["rms_rel"] - 1e-7 / np.sqrt(2)) < 1e-12 and abs(e["cost_rel"] - 1e-7) < 1e-12
    u64 = np.full((3, 2), 0.5)
    assert R.solve_error(u64 + 0.9e-4, u64)["within"] and not R.solve_error(u64 + 2.1e-4, u64)["within"]
    t = R.summarize({"a": [e, e], "b": [dict(e, rms_rel=2 * e["rms_rel"], max_rel=3 * e["max_rel"])] * 2}, {"a": [R.solve_error(u64, u64)] * 2, "b": [R.solve_error(u64 + 1e-3, u64)] * 2})
    assert t["a"]["solves_within_1e-4_of_float64"] == 1.0 and t["b"]["solves_within_1e-4_of_float64"] == 0.0
    r = R.ratios_to(t, base="a")
    assert abs(r["b"]["grad_rms_rel"] - 2.0) < 1e-12 and abs(r["b"]["grad_max_rel_worst"] - 3.0) < 1e-12
    assert R.run_threads([lambda i=i: i * i for i in range(7)], 3) == [i * i for i in range(7)]
