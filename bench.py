#!/usr/bin/env python3
"""Benchmark of the MPC inner loop on MI355X: MPC solves/sec (+ p50 solve latency), Iris H=50 P=128.

Contract: `python bench.py --gpus N --steps K --warmup W`. For N>1 either launch it under torch.distributed.run
(one rank per GPU over RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or call it plainly:
without WORLD_SIZE in the environment the parent — before anything touches the GPU — starts the N ranks itself as
child processes (rendezvous on 127.0.0.1), relays rank 0's JSON line and exits with the children's return code.
A "step" = one launch of the solve kernel over one batch of B independent MPC problem instances per GPU
(synthetic initial states; inputs already resident in HBM). Weak scaling: B per GPU is fixed, instances are
sharded one batch per GPU with no data-path collective; rank 0 broadcasts the shared model blob once over RCCL
at start-up. Rank 0 prints ONE JSON line. The timed launch's outputs of sampled instances are compared bit for
bit with the CPU oracle solving the same instances (verified_instances / verified_bit_exact in the line).

This file: argument parsing, the timed leg, the JSON line. The pieces live in benchlib/ (counts, verify, power, ranks, legs).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.counts import F32_MFMA_PEAK_TF, HBM_PEAK_GBS, checkpoint_bytes, roofline_of      # noqa: E402
from benchlib.counts import F32_SCALAR_VALU_PEAK_TF      # noqa: E402
from benchlib.ranks import require_same_on_every_rank, spawn_ranks      # noqa: E402
from benchlib.verify import Verifier, calibrate_checker_seconds, cpu_c1_single_solve_ms, effective_cores, instruction_model_check, sample_indices      # noqa: E402

_T0 = time.perf_counter()
_REAL_STDOUT = None


def own_stdout():
    """The driver parses ONE JSON line from rank 0's stdout, and libraries below us write there too (gloo: "[Gloo] Rank 0 is connected to ..."):
    from here on file descriptor 1 points at stderr and only emit() writes to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def progress(msg):
    """one line on stderr per phase: a default run takes minutes and must not look hung to whoever started it"""
    print(f"[bench {time.perf_counter() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)


def lib_hash():
    """sha256 (first 16 hex digits) of the library the solver loaded: ties profiles/pmc_traffic.json to the build it was measured on"""
    import hashlib
    from sde4mbrl_px4_amd import _abi
    return hashlib.sha256(open(_abi.lib_path(), "rb").read()).hexdigest()[:16]


DEFAULT_MLP = "f32x3"      # the reported arithmetic: f32 operands and accumulation, layer-2 contractions as three-limb bf16 splits (SPEC.md 9b) ...
DEFAULT_MATH = "fast"      # ... and tanh / sigmoid / rsqrt on the hardware's transcendental instructions (SPEC.md 10), which the oracle evaluates
                           # through its exact model of v_exp_f32 / v_rcp_f32 / v_rsq_f32 (SPEC.md 10a): every figure of the line is checked bit for bit

DTYPE_NOTE = {
    "f32": "f32",
    "f32x3": "f32 (state, accumulation and every operand f32; the two 32x32 MLP contractions per step evaluated as three-limb bf16 splits of "
             "both f32 operands on v_mfma_f32_32x32x16_bf16: error against float64 not larger than the f32 fma chain's "
             "(tests/test_mfma16_model_cpu.py), bit-identical to the CPU oracle)",
    "f16": "f16 MLP operands / f32 accumulate and state",
}
MATH_NOTE = {"exact": "activations by the software forms of SPEC.md 3", "fast": "activations on v_exp_f32 / v_rcp_f32 (1 ulp), kept as r = 1 / (1 + 2^a') with their affine maps folded into the weights (SPEC.md 10b), quaternion "
             "renormalisation on v_rsq_f32; in f32x3 the forward layer-2 contraction takes two binary16 limbs (round to nearest, 2^-24) of the bounded activations and of the "
             "weights on v_mfma_f32_32x32x16_f16 (SPEC.md 10c), and so do all four contractions of the adjoint sweep behind a per-particle power-of-two scale (SPEC.md 10e; "
             "mlp_dtype f16 too); bit-identical to the CPU oracle through its models of the three "
             "transcendental instructions (SPEC.md 10a) and of the matrix instruction (9a)"}


def dry_run(args, rank, world):
    """SDEMPC_BENCH_DRY=1 (tests, no GPU): the rank plumbing of an N-rank run — rendezvous, blob broadcast, all-reduce MAX, barriers — over gloo on
    the CPU, then one JSON line from rank 0. SDEMPC_BENCH_FAIL_RANK=r makes rank r fail before the rendezvous (what a rank without a usable GPU does)."""
    import torch.distributed as dist
    from sde4mbrl_px4_amd import synthetic_iris
    from sde4mbrl_px4_amd.dist import broadcast_blob, max_over_ranks, max_over_ranks_each
    if os.environ.get("SDEMPC_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(f"bench.py: rank {rank}: simulated start-up failure (SDEMPC_BENCH_FAIL_RANK)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = synthetic_iris().to_blob() if rank == 0 else b""
    blob = broadcast_blob(blob, src=0)
    blob = corrupted_for_test(blob, rank)
    same = require_same_on_every_rank({"model blob": blob, "library build": "dry run"})
    dist.barrier()
    slowest = max_over_ranks(1.0 + rank)
    ticks = max_over_ranks_each([float(rank), float("inf") if rank == world - 1 else 0.5])
    # every rank proves its own results: here a two-iteration C1-sized solve "of the GPU" (the oracle stands in for it: there is no GPU in a dry run)
    # checked by the rank's own verifier, then the collectives of the real run (benchlib/verify.py, all_ranks_verified)
    from benchlib.verify import stand_in_outputs
    from sde4mbrl_px4_amd import load_mpc_config, prng
    from sde4mbrl_px4_amd import workload as W
    cfg = load_mpc_config(os.path.join(ROOT, "configs", "c1_iris_posctrl_h20_p32.yaml")).replace(max_iter=2, max_no_improvement_iter=2)
    n = 2
    x0 = W.random_initial_states(n, rank * n)
    xref = np.stack([W.constant_reference(W.HOVER, cfg.horizon)] * n)
    keys = prng.split(prng.PRNGKey(10), world * n)[rank * n:(rank + 1) * n]
    u0 = np.tile(np.asarray(cfg.uref, np.float32)[None, None], (n, cfg.horizon, 1))
    got = stand_in_outputs(cfg, blob, x0, xref, keys, u0, cfg.ls_init_stepsize)
    if os.environ.get("SDEMPC_BENCH_CORRUPT_OUTPUT_RANK") == str(rank):
        got[0].view(np.uint32)[1, 3, 2] ^= 1          # one bit of one control of this rank's last instance
    V = Verifier(1)
    V.add("main", cfg, blob, [0, n - 1], x0, xref, keys, u0, cfg.ls_init_stepsize, got)
    V.start(); V.join()
    by_rank, bad_total = all_ranks_verified(V, rank, world)
    dist.barrier()
    if rank == 0:
        emit({"dry_run": True, "n_gpus": world, "blob_bytes": len(blob), "max_over_ranks": slowest, "ticks": [t if np.isfinite(t) else None for t in ticks],
              "same_on_every_rank": sorted(same), "verified_by_rank": by_rank, "bad_words_all_ranks": bad_total})
    dist.destroy_process_group()
    bad_ranks = [r for r, (ok, asked) in enumerate(by_rank) if ok != asked]
    if bad_total or bad_ranks:
        raise SystemExit(f"bench.py: rank {rank}: the outputs of rank(s) {bad_ranks} differ from the oracle ({bad_total} words in all; (checked ok, asked) by rank: {by_rank})")


def corrupted_for_test(blob, rank):
    """test hook SDEMPC_BENCH_CORRUPT_RANK=r: rank r's copy of the broadcast model blob gets one bit flipped (what a faulty link or a stale file
    would do): the cross-rank fingerprint check must stop the run and name the rank"""
    if os.environ.get("SDEMPC_BENCH_CORRUPT_RANK") == str(rank):
        b = bytearray(blob); b[len(b) // 2] ^= 0x10
        return bytes(b)
    return blob


def all_ranks_verified(V, rank, world, device=None, force=False):
    """Every rank has checked instances of ITS OWN launches (V: its verifier, drained). Returns ([(checked ok, asked) per rank], words differing in
    the whole job) on every rank: SUM of the bad words, all-gather of the per-rank counts — a wrong device binding or corrupted weights on rank 5
    cannot hide behind rank 0's clean sample."""
    from sde4mbrl_px4_amd.dist import gather_int64, sum_over_ranks
    asked = sum(len(r["idx"]) for r in V.results.values())
    bad = sum(r["bad_words"] for r in V.results.values())
    ok = sum(r["done"] for r in V.results.values()) if bad == 0 else sum(r["done"] for r in V.results.values() if r["bad_words"] == 0)
    rows = gather_int64([ok, asked], device=device, force=force)
    total = sum_over_ranks([bad], device=device, force=force)[0]
    return [tuple(r) for r in rows], total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=12288, help="MPC problem instances per GPU per step (eight rounds of the 1,536 instances an MI355X holds "
                    "at once: 256 CUs x 3 workgroups x 2 instances; from three rounds on the persistent launch hands instances out by ticket)")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--latency-reps", type=int, default=1000)
    ap.add_argument("--latency-warmup", type=int, default=20)
    ap.add_argument("--mlp-dtype", default=DEFAULT_MLP, choices=["f32", "f16", "f32x3"],
                    help="f32x3 (default, the reported metric): f32 arithmetic with the layer-2 contractions as three-limb bf16 splits on the matrix pipe "
                         "(SPEC.md 9b), bit-identical to the oracle; f32: every contraction an f32 fma chain; f16: fp16-operand MLP contractions (SPEC.md 9)")
    ap.add_argument("--math-mode", default=DEFAULT_MATH, choices=["exact", "fast"],
                    help="fast (default): hardware transcendentals (SPEC.md 10), checked bit for bit through the oracle's instruction model (10a); exact: the software forms of SPEC.md 3")
    ap.add_argument("--max-iter", type=int, default=0, help="override the YAML's apg_mpc.max_iter (0: keep; profiling runs of the long-horizon config)")
    ap.add_argument("--no-other-math-mode", "--no-tolerance-modes", dest="no_other_math_mode", action="store_true", help="skip the extra launches in the other math mode (exact <-> fast)")
    ap.add_argument("--verify-budget-s", type=float, default=90.0, help="wall seconds the bit-exact checks of the timed launch may take on this host's cores: sets how many "
                    "instances are checked (at least 6, at most 24; a three-iteration solve calibrates the checker's speed)")
    ap.add_argument("--referee", type=int, default=16, help="instances of the timed batch solved by the float64 build of the oracle for the vs_float64 table (0: none)")
    ap.add_argument("--c4-reps", type=int, default=100, help="N > 1: barrier-aligned ticks of the one-instance-per-GPU leg (BASELINE config 4); 0 skips it")
    ap.add_argument("--no-power", action="store_true", help="do not sample package power / shader clock beside the timed launches")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary legs (C2 f32 chain, C3, C5 f32 / f16)")
    ap.add_argument("--verify", type=int, default=-1, help="instances of the timed launch checked bit for bit against the CPU oracle "
                    "(-1: a sample across the batch incl. ticket-drawn instances; 0: none)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-start: this process never touches the GPU (no torch.cuda / HIP call has been made yet)
        raise SystemExit(spawn_ranks(args.gpus, __file__, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    own_stdout()
    if os.environ.get("SDEMPC_BENCH_DRY") == "1":
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    from benchlib.legs import Leg, config4_leg, float64_referee_leg, latency_of, other_config_legs, other_math_mode_legs
    from benchlib.power import PowerSampler
    from sde4mbrl_px4_amd import load_mpc_config, synthetic_hexa, synthetic_iris

    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py: rank {rank} needs a GPU: the product path has no CPU fallback")
    # test hooks (single-GPU boxes): SDEMPC_BENCH_DEVICE pins every rank to one ordinal, SDEMPC_BENCH_BACKEND=gloo replaces RCCL,
    # SDEMPC_BENCH_FORCE_DIST=1 initialises the process group even for one rank (the RCCL branch — init, device-tensor broadcast, all-reduce,
    # barrier, destroy — then runs on a one-GPU box); the driver's multi-GPU runs use none of them (one GPU per rank, backend nccl = RCCL over xGMI)
    dev_ord = int(os.environ.get("SDEMPC_BENCH_DEVICE", local_rank))
    backend = os.environ.get("SDEMPC_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("SDEMPC_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(dev_ord)
    dev = torch.device("cuda", dev_ord)
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:      # named on this rank's stderr; the self-starting parent quotes it in its summary line (benchlib/ranks.py)
            raise SystemExit(f"bench.py: rank {rank} of {world}: init_process_group({backend!r}) on device {dev_ord} failed: {type(e).__name__}: {e}")

    def cfg_of(path, mlp, **kw):
        return load_mpc_config(path).replace(**dict({"mlp_dtype": mlp, "math_mode": args.math_mode}, **kw))

    cfg = cfg_of(args.config, args.mlp_dtype)
    if args.max_iter:
        cfg = cfg.replace(max_iter=args.max_iter, max_no_improvement_iter=args.max_iter)
    H, P, m, B = cfg.horizon, cfg.num_particles, cfg.num_motors, args.batch
    c2_run = os.path.basename(args.config).startswith("c2_")
    # shared model: rank 0 builds it, RCCL broadcast over xGMI (read-only weights are the only shared data)
    from sde4mbrl_px4_amd.dist import broadcast_blob, max_over_ranks
    blob = (synthetic_iris() if m == 4 else synthetic_hexa()).to_blob() if rank == 0 else b""
    blob = corrupted_for_test(broadcast_blob(blob, src=0, device=dev, force=force_dist), rank)
    if use_dist:        # every rank holds the same weights and runs the same build of the library, or the run stops here naming the rank that does not
        require_same_on_every_rank({"model blob": blob, "library build": lib_hash()}, device=dev, force=force_dist)
    if args.math_mode == "fast" and args.verify != 0:
        instruction_model_check(ROOT, rank, progress)

    L = Leg(cfg, blob, B, dev_ord, rank, world, pos="posctrl" in os.path.basename(args.config))

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    sampler = None
    if not args.no_power and rank == 0:        # rank 0 samples every GPU of the job (sysfs; no child process, no collective)
        ords = [int(os.environ["SDEMPC_BENCH_DEVICE"])] * world if "SDEMPC_BENCH_DEVICE" in os.environ else list(range(world))
        pci = {}
        for o in set(ords):
            try:
                pr = torch.cuda.get_device_properties(o)
                pci[o] = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            except Exception:
                pass
        sampler = PowerSampler(ords, pci)
        sampler.start()
    if rank == 0:
        progress(f"timed leg: {args.warmup} warm-up + {args.steps} timed launches of {B} instances per rank")
    for _ in range(args.warmup):
        L.step()
    sync_all()
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        L.step()
        if rank == 0 and (i_step + 1) % 8 == 0:
            progress(f"launch {i_step + 1} of {args.steps} enqueued")      # (asynchronous: no sync inside the timed region)
    sync_all()
    t1 = time.perf_counter()
    power = sampler.summary(t0, t1) if sampler else None
    if rank == 0:
        progress(f"timed leg done: {t1 - t0:.1f} s" + (f"; package {power['package_power_w_median']} W of {power['power_cap_w']}, sclk {power['sclk_mhz_median']} MHz" if power else ""))
    elapsed = max_over_ranks(t1 - t0, device=dev, force=force_dist)
    # per-launch kernel duration from HIP events, measured live on the launch stream (separate launches)
    ev_ms, n_grad, n_fwd = L.timed_events(min(args.steps, 3))
    kernel_name = L.solver.last_kernel_name()       # the instantiation the timed launches ran, as rocprofv3 names it
    uopt_h, xevol_h, info_h = L.host_outputs()        # outputs of the timed configuration (same inputs, deterministic kernel)
    n_it = float(info_h[:, 2].mean())
    n_ls = float(info_h[:, 7].mean())

    c4, c4_check = None, None
    if use_dist and args.c4_reps > 0:
        if rank == 0:
            progress(f"config 4: one instance per GPU, {args.c4_reps} barrier-aligned ticks")
        c4, c4_check = config4_leg(L, cfg, blob, dev, dev_ord, world, args.c4_reps, args.mlp_dtype, sync_all, force_dist)

    out, lat_done = None, None
    if rank == 0:
        solves = world * B * args.steps
        value = solves / elapsed
        k_ms = float(np.mean(ev_ms))
        ach_tf, ach_gbs, bytes_solve, b_grad, b_ls = roofline_of(cfg, B, k_ms, n_grad, n_fwd)
        # HBM bytes per launch and the vector-issue figures from the PMC counters: they need rocprofv3 around the process (separate --pmc passes,
        # MI355X_MICROARCH.md), so they come from the committed summary of those passes for this config, batch, arithmetic and BUILD
        # (tools/profile_round.sh -> profiles/pmc_traffic.json records the sha256 of the library it measured): null when the loaded library is
        # another build than the one the counters were taken on
        traffic, traffic_src, traffic_build, valu_issue = None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        build = lib_hash()
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(f"{os.path.basename(args.config)}:B{B}:{args.mlp_dtype}:{args.math_mode}", {})
                traffic_build = rec.get("traffic_build")
                if traffic_build == build:
                    traffic, traffic_src, valu_issue = rec.get("hbm_bytes_per_launch"), rec.get("source"), rec.get("valu_issue")
                else:
                    traffic_src = f"stale: counters were taken on build {traffic_build}, this run loaded {build}"
            except Exception:
                traffic, traffic_src = None, None
        # p50 / p95 latency of a single solve (B=1 launches), outside the timed region: SURVEY.md §8(d) protocol, >= 20 warm-up
        # and >= 1000 timed solves by default. A single instance is a latency problem, not a throughput one: the library's latency layouts
        # (one particle per wave over ceil(P/4) x 7 workgroups, speculative line search) exist for the f32 fma-chain arithmetic, so the p50
        # is measured on a handle in mlp_dtype f32 — what a deployment that cares about one vehicle's tick would configure — and the
        # single-instance time in the arithmetic of the throughput number (tile layout on one workgroup) is reported beside it.
        from sde4mbrl_px4_amd.solver import SdeMpcSolver
        progress(f"single-solve latency: {args.latency_reps} solves")
        if args.mlp_dtype == "f32":
            lat, single_kernel, fallbacks = latency_of(L, L.solver, args.latency_reps, args.latency_warmup)
            lat_same, same_kernel = lat, single_kernel
        else:
            s_lat = SdeMpcSolver(cfg.replace(mlp_dtype="f32"), blob, max_batch=8, device=dev_ord)
            lat, single_kernel, fallbacks = latency_of(L, s_lat, args.latency_reps, args.latency_warmup)
            s_lat.close()
            lat_same, same_kernel, _ = latency_of(L, L.solver, min(args.latency_reps, 30), min(args.latency_warmup, 2))
        pw = (power or {}).get("package_power_w_median")
        energy = (pw * (elapsed / args.steps) / B) if isinstance(pw, (int, float)) else None       # per GPU: this GPU's watts x its launch time / its instances
        out = {
            "metric": (f"MPC solves/sec, Iris H=50 P=128, arithmetic {args.mlp_dtype}/{args.math_mode} (p50 solve latency in p50_solve_latency_ms: arithmetic f32/{args.math_mode}; "
                       "same-arithmetic pairs in by_arithmetic; every arithmetic against float64 in vs_float64)" if c2_run else f"MPC solves/sec, {os.path.basename(args.config)}, arithmetic {args.mlp_dtype}/{args.math_mode}"),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NOTE[args.mlp_dtype] + "; " + MATH_NOTE[args.math_mode], "mlp_dtype": args.mlp_dtype, "math_mode": args.math_mode, "data": "synthetic",
            "config": {"workload": f"{os.path.basename(args.config)}: H={H} P={P} m={m}, {B} independent MPC instances per GPU per step, "
                                   f"cold-start solves from the hover guess, max_iter={cfg.max_iter} maxls={cfg.ls_maxls}",
                       "instances_per_gpu": B, "noise": "threefry2x32 keys (seed 10 split per instance), normal draws generated on the device", "N_it_mean": n_it, "N_ls_mean": n_ls, "N_grad_evaluated_mean": n_grad, "N_forward_rollouts_mean": n_fwd, "parallelism": f"instances sharded over {world} GPU(s), no data-path collective"},
            "p50_solve_latency_ms": float(np.median(lat)) if lat else None,
            "p95_solve_latency_ms": float(np.percentile(lat, 95)) if lat else None,
            "latency_reps": len(lat), "latency_layout_fallbacks": fallbacks, "latency_kernel": single_kernel, "latency_mlp_dtype": "f32", "latency_math_mode": args.math_mode,
            "p50_solve_latency_ms_in_the_throughput_arithmetic": float(np.median(lat_same)) if lat_same else None,
            "latency_kernel_in_the_throughput_arithmetic": same_kernel, "latency_reps_in_the_throughput_arithmetic": len(lat_same),
            "p50_solve_latency_note": "one instance alone on the GPU (B = 1 launch of the same C-ABI entry point) on a handle in mlp_dtype f32: the library spreads it over ceil(P/4) x 7 "
                                      "workgroups (one particle per wave; three line-search trials and the candidate gradients of the next iteration evaluated at once). The latency layouts "
                                      "have no matrix instructions, so they exist for the f32 fma-chain arithmetic only; a single instance in the f32x3 arithmetic runs in the tile layout on one "
                                      "workgroup (the second figure). Both arithmetics are bit-identical to the oracle in their mode and agree with each other to 1e-6 on the controls",
            "p50_batch_latency_ms": float(np.median(ev_ms)),
            "library_build": build,
            "power": power,
            "energy_per_solve_J": energy,
            "energy_per_solve_note": "median package power of this GPU over the timed launches x ms_per_step / instances per GPU: at the power cap this, not the clock, is what a faster kernel must lower",
            "c4_one_instance_per_gpu": c4,
            "roofline": {"bound": "valu-issue @ power cap", "achieved": ach_tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach_tf / F32_MFMA_PEAK_TF,
                         "peak_scalar_valu_tflops": F32_SCALAR_VALU_PEAK_TF, "frac_of_scalar_valu_peak": ach_tf / F32_SCALAR_VALU_PEAK_TF,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_build": traffic_build, "kernel": kernel_name, "kernel_ms": k_ms,
                         "binding_resource": ("the package power cap (field power: the firmware holds the shader clock below 2.4 GHz) and, inside it, what the three waves of a SIMD wait for: vector (VALU) instruction "
                                              "issue with its LDS / matrix-chain dependencies"
                                              + ((" — PMC passes of this build: the vector stream alone is %.0f %% of the launch's cycles at the best rate a SIMD issues this mix, the matrix pipe is busy %.0f %% of them"
                                                  % (100 * valu_issue["frac_at_best_issue"], 100 * valu_issue["mfma_busy_frac"])) if valu_issue else "")
                                              + "; per wave half of the cycles are issue stalls (dependent instructions, arbitration among the three waves), 15 % waits for LDS / memory operands; HBM moves 3.1 of 8 TB/s "
                                                "(profiles/r5_c2_pmc.json). Round 5 showed that the COUNT of vector instructions is not what binds (profiles/r5_ab.txt 2)"),
                         "valu_issue": valu_issue,
                         "note": "algorithmic flops = SURVEY 8d MLP formula x P*H*(2*N_grad+N_ls+2) (N_grad = gradient evaluations actually performed: sdempc_work_counters). `frac` keeps the series of "
                                 "the earlier rounds: against the 157.3 TFLOP/s f32 peak (f32 vector peak = f32-input MFMA peak), which the vector ALUs reach only with packed-f32 instructions throughout — "
                                 "measured as a loss in this kernel; frac_of_scalar_valu_peak is the same flops against what one fma per lane and cycle gives (78.6 TFLOP/s at 2.4 GHz). Neither unit is what binds: "
                                 "`bound` names it — vector instruction ISSUE (not flops: activations, rigid body, adjoint algebra, limb splits count as instructions, few as flops) under the package "
                                 "power cap (binding_resource / valu_issue: vector instructions per SIMD, shader cycles, and the fraction of the cycles the stream takes at the best rate a SIMD issues it; PMC passes "
                                 "of the same build, profiles/). In the f32x3 mode 74 % of the counted flops (the two 32x32 contractions) run as limb products on the matrix pipe (2.5 PFLOP/s dense): six bf16 "
                                 "products per contraction, in math_mode fast four binary16 ones in the forward sweeps (SPEC.md 9b, 10c)"},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                             "traffic": traffic, "bytes_per_solve": bytes_solve, "B_grad": b_grad, "B_ls": b_ls,
                             "checkpoint_bytes_per_solve": checkpoint_bytes(cfg, n_grad),
                             "note": "achieved = SURVEY 8d algorithmic bytes / kernel time; measured traffic additionally contains the activation-checkpoint stream"},
        }
        if world > 1:
            out["cpu_baseline"] = "skipped (n_gpus > 1): reported by the single-GPU run"
    if use_dist:
        dist.barrier()         # (rank 0's latency loop is over: the other ranks' checker threads must not compete with its host timestamps)

    # ---- every rank proves its own results: instances of ITS timed launch (and of its config-4 ticks) against the CPU oracle, on its share of the
    # host cores; the other GPU legs of a single-GPU run go on meanwhile ------------------------------------------------------------------------
    cores = args.cpu_threads or min(effective_cores(), 64)
    nthr = max(1, cores // max(world, 1))
    V = Verifier(max(1, nthr - 1) if world == 1 else nthr)
    others, plan, disputed = {}, {}, None
    if args.verify != 0:
        # how many full-length instances fit the wall-time budget on this host: the checker is timed on a three-iteration solve of instance 0
        est_s, cal_s = calibrate_checker_seconds(cfg, blob, L.x0_h[0], L.xref_h[0], L.keys[0], L.u0_h[0], L.s0, n_it, n_ls)
        fit = int(0.7 * args.verify_budget_s * V.n_threads / max(est_s, 1e-3))
        if args.verify > 0:
            idx = list(range(min(args.verify, B)))
        elif world > 1:
            idx = sample_indices(B, L.slots(), n_initial=1, n_drawn=max(0, min(fit, 6) - 2))       # first, last, ticket-drawn ones as they fit: every rank at least 2
        else:
            n = max(6, min(24, fit))            # never fewer than six: the teams' first assignments, ticket-drawn ones across the batch, the last
            idx = sample_indices(B, L.slots(), n_initial=max(2, n // 4), n_drawn=n - max(2, n // 4) - 1)
        plan = {"budget_s": args.verify_budget_s, "threads": V.n_threads, "estimated_cpu_s_per_instance": est_s, "calibration_s": cal_s, "instances_that_fit": fit, "asked": len(idx)}
        V.add("main", cfg, blob, idx, L.x0_h, L.xref_h, L.keys, L.u0_h, L.s0, (uopt_h, xevol_h, info_h))
        if c4_check is not None:
            i4 = c4_check["idx"]
            V.add("c4", c4_check["cfg"], blob, [0], [L.x0_h[i4]], [L.xref_h[i4]], [L.keys[i4]], [L.u0_h[i4]], L.s0, tuple(a[None] for a in c4_check["got"]))
        # float64 referee: the first instances of the timed batch by the float64 build of the oracle (vs_float64; rank 0 of a single-GPU run)
        n_ref = min(args.referee, B) if (rank == 0 and world == 1 and c2_run and not args.max_iter) else 0
        ug = None
        if n_ref:
            ug = np.clip(L.u0_h[:n_ref] + 0.1 * np.random.default_rng(7).standard_normal(L.u0_h[:n_ref].shape), 1e-4, 1).astype(np.float32)
            V.add_referee(cfg, blob, range(n_ref), L.x0_h, L.xref_h, L.keys, L.u0_h, ug, L.s0)
        V.start()
        if rank == 0:
            progress(f"CPU verification of {len(idx)} instances of the timed launch started on {V.n_threads} threads (estimated {est_s:.0f} CPU-s each; {fit} would fit {args.verify_budget_s:.0f} s)")
    else:
        n_ref, ug = 0, None
    if rank == 0 and world == 1:
        if not args.no_other_configs and c2_run and not args.max_iter:
            others = other_config_legs(ROOT, args.mlp_dtype, args.math_mode, B, dev, dev_ord, cfg_of, V, args.verify, progress)
        if not args.no_other_math_mode and c2_run and not args.max_iter:
            out["other_math_mode"], disputed = other_math_mode_legs(L, cfg, blob, dev_ord, uopt_h)
            if disputed and args.referee and args.verify != 0:
                V.add_referee(cfg, blob, [i for i in disputed["idx"] if i >= n_ref], L.x0_h, L.xref_h, L.keys, L.u0_h, None, L.s0)
        if n_ref:
            progress(f"float64 referee: the first {n_ref} instances in the four f32 arithmetics on the GPU")
            gpu_ref = float64_referee_leg(L, cfg, blob, dev_ord, n_ref, ug)
        # the two same-arithmetic pairs of the headline configuration side by side (a handle has ONE mlp_dtype)
        if c2_run and not args.max_iter:
            p50 = out["p50_solve_latency_ms"]
            mm = args.math_mode
            by = {f"{args.mlp_dtype}/{mm}": {"solves_per_s": out["value"], "p50_ms": out["p50_solve_latency_ms_in_the_throughput_arithmetic"],
                                             "layouts": "persistent duo tiles / " + ("speculative one-particle-per-wave" if args.mlp_dtype == "f32" else "tile layout on one workgroup")}}
            if args.mlp_dtype != "f32":
                by[f"f32/{mm}"] = {"solves_per_s": others.get("c2_f32_chain", {}).get("value"), "p50_ms": p50, "layouts": "persistent duo tiles / speculative one-particle-per-wave"}
            om = out.get("other_math_mode", {})
            for k, v in om.items():
                if isinstance(v, dict):
                    by[k] = {"solves_per_s": v.get("value"), "p50_ms": v.get("p50_ms"), "layouts": "persistent duo tiles / " + ("speculative one-particle-per-wave" if v.get("p50_ms") else "-")}
            out["by_arithmetic"] = by
        # reported CPU baseline: the particle-vectorised build (f32 fma-chain arithmetic), one solve at a time per thread, on instances of the same
        # workload (the first instances of the GPU batch); queued behind the checks, it fills the threads their drain leaves idle
        do_cpu = not args.no_cpu_baseline
        n_cpu = min(40 * V.n_threads, B) if do_cpu else 0
        cfg32 = cfg.replace(mlp_dtype="f32", math_mode="exact")      # (the timing build has the f32 fma chains and the software activations of SPEC.md 3 only)
        if do_cpu:
            if not V.threads:
                V.start()
            V.add_baseline(cfg32, blob, 40, L.x0_h[:n_cpu], L.xref_h[:n_cpu], L.keys[:n_cpu], L.u0_h[:n_cpu], L.s0)
        progress("waiting for the CPU threads (checks of the timed launches, float64 referee, CPU baseline)")
    wall = V.join()
    by_rank, bad_total = all_ranks_verified(V, rank, world, device=dev, force=force_dist)
    bad_ranks = [r for r, (ok, asked) in enumerate(by_rank) if ok != asked]
    short = V.incomplete()
    if rank == 0:
        for leg, r in V.results.items():
            ok = r["bad_words"] == 0 and r["done"] == len(r["idx"])
            tgt = out if leg == "main" else (out["c4_one_instance_per_gpu"] if leg == "c4" else others[leg])
            if isinstance(tgt, dict):
                tgt["verified_instances"] = r["done"]
                tgt["verified_indices"] = r["idx"] if leg != "c4" else [c4_check["idx"]]
                tgt["verified_bit_exact"] = ok
                if "committed" in r:
                    tgt["verified_against"] = "tests/golden/" + r["committed"]
        out["verified_by_rank"] = [{"rank": r, "checked_bit_exact": ok, "asked": asked} for r, (ok, asked) in enumerate(by_rank)]
        out["verification_plan"] = plan
        if "main" in V.results:
            r = V.results["main"]
            out["verified_note"] = ("uopt, xevol and the 8 telemetry words of instances %s of the timed launch (the teams' initial assignments are b < %d; the others were handed "
                                    "out by ticket) compared bit for bit with the CPU oracle (oracle/sde_mpc_oracle.c, mlp_dtype %s through oracle/mfma16_model.c, math_mode %s) solving the "
                                    "same instances from the same keys; %d words differ; oracle time %.0f s on %d threads beside the GPU legs; on N > 1 GPUs EVERY rank checks instances of "
                                    "its own launches the same way (verified_by_rank) and the ranks' model blobs and library builds are fingerprinted against each other at start-up"
                                    % (r["idx"], L.slots(), args.mlp_dtype, args.math_mode, r["bad_words"], r["cpu_s"], V.n_threads))
        else:
            out["verified_instances"], out["verified_bit_exact"] = 0, None
        if others:
            out["other_configs"] = others
        if n_ref and all(i in V.referee and V.referee[i][0] is not None for i in range(n_ref)):
            out["vs_float64"] = vs_float64_table(V.referee, gpu_ref, n_ref, ug is not None)
        if disputed and all(i in V.referee for i in disputed["idx"]):
            this = f"{args.mlp_dtype}/{args.math_mode}"
            rows = []
            for k, i in enumerate(disputed["idx"]):
                u64 = V.referee[i][2].astype(np.float64)
                rows.append({"instance": i, "max_abs_du_between_the_modes": float(np.abs(uopt_h[i].astype(np.float64) - disputed["u_other"][k]).max()),
                             f"max_abs_du_{this}_vs_float64": float(np.abs(uopt_h[i] - u64).max()), f"max_abs_du_{disputed['other']}_vs_float64": float(np.abs(disputed["u_other"][k] - u64).max())})
            a = np.array([r[f"max_abs_du_{this}_vs_float64"] for r in rows]); b = np.array([r[f"max_abs_du_{disputed['other']}_vs_float64"] for r in rows])
            out["other_math_mode"]["disputed_instances_vs_float64"] = {
                "instances": rows, f"{this}_closer_to_float64_in": int((a < b).sum()), f"{disputed['other']}_closer_to_float64_in": int((b < a).sum()),
                f"{this}_within_1e-4_of_float64_in": int((a <= 2e-4).sum()), f"{disputed['other']}_within_1e-4_of_float64_in": int((b <= 2e-4).sum()),
                "note": "the eight instances of the timed batch whose controls differ most between the two math modes, each against the float64 solve of the same instance: when a "
                        "rounding flips a line-search decision the float64 evaluation takes one of the two branches — whichever f32 arithmetic took the same one stays close, the other does not; "
                        "neither mode is systematically the closer one (counts; 'within' uses abs + rel 1e-4 as 2e-4 on controls in [0, 1])"}
        br = V.baseline_rate()
        if world == 1 and br is not None and not (bad_total or short):
            v, threads_used, n_done, busy = br
            outs_f = V.baseline["outs"]
            devs = np.array([float(np.max(np.abs(outs_f[i][0] - uopt_h[i]))) for i in sorted(outs_f)])
            md, mmed = float(devs.max()), float(np.median(devs))
            c1, c1cfg = cpu_c1_single_solve_ms(blob)
            rmain = V.results.get("main")
            out["cpu_baseline"] = {"value": v, "unit": "solves/s", "cores": threads_used, "kind": "port",
                                   "threads_used": threads_used, "os_cpu_count": os.cpu_count(), "usable_cores": effective_cores(),
                                   "sample": f"{n_done} solves of the same workload (the first {n_done} instances of the GPU batch, one solve at a time per thread, "
                                             f"{threads_used} threads of the {effective_cores()} usable host cores: os.cpu_count {os.cpu_count()}; the busiest thread spent {busy:.1f} s on them) "
                                             "by the particle-vectorised build of the C oracle (oracle/sde_mpc_oracle.c -DORC_VEC: 16 particles per call, -O3 -march=native, "
                                             "contraction allowed, f32 fma-chain contractions, software activations of SPEC.md 3; CPU restatement of SPEC.md, not the reference JAX path: that cannot run here). "
                                             "value = sum over the threads of (solves of the thread / the time it spent on them): every worker thread solves its own 40 instances once the queue of "
                                             "bit-exact checks has nothing left for it, so the cores are busy with checks or baseline solves throughout (the tail of the last threads excepted); "
                                             f"|uopt - GPU uopt| over the sample (200-iteration solves; timing build, not the checker): median {mmed:.1e}, max {md:.1e}",
                                   "value_bit_exact_build": (rmain["done"] / rmain["cpu_s"] * 1.0) if rmain and rmain["cpu_s"] > 0 else None,
                                   "value_bit_exact_build_note": "solves per second PER THREAD of the bit-exact checker in the arithmetic of this run (the matrix-instruction model is integer code)",
                                   "cpu_c1_single_solve_ms": c1["vec"][0], "cpu_c1_single_solve_ms_scalar_build": c1["scalar"][0],
                                   "cpu_c1_note": f"BASELINE config 1: c1_iris_posctrl_h20_p32.yaml H={c1cfg.horizon} P={c1cfg.num_particles}, one cold-start solve "
                                                  f"(N_it {c1['vec'][1]:.0f}) on ONE thread, median of 3: particle-vectorised build / bit-exact scalar -O2 build"}
        if short or V.errors:
            out["verification_errors"] = V.errors[:8]
        out["bench_wall_s"] = time.perf_counter() - _T0
        progress(f"done: {out['bench_wall_s']:.0f} s in all; CPU threads busy for {wall:.0f} s")
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    L.close()
    if bad_total or bad_ranks or short:
        raise SystemExit(f"bench.py: rank {rank}: outputs of timed launches differ from the oracle: rank(s) {bad_ranks}, {bad_total} words in all ((checked ok, asked) by rank: {by_rank})" if (bad_total or bad_ranks) else
                         f"bench.py: rank {rank}: the verification did not complete (checked, asked) by leg: {short}; first errors: {V.errors[:3]}")
    return 0


def vs_float64_table(ref64, gpu, n, with_grad=True):
    """the four f32 arithmetics as the GPU evaluated them (bit-identical to their oracles) against the float64 oracle on the first n instances of the timed batch"""
    from benchlib import referee as R
    grad_rows, solve_rows = {}, {}
    for k, (g, c, u) in gpu.items():
        grad_rows[k] = [R.gradient_error(g[i], ref64[i][0], c[i], ref64[i][1]) for i in range(n)]
        solve_rows[k] = [R.solve_error(u[i], ref64[i][2]) for i in range(n)]
    table = R.summarize(grad_rows, solve_rows)
    return {"instances": n, "by_arithmetic": table, "per_gradient_ratio_to_f32_exact": R.ratios_to(table),
            "float64_cpu_s_per_instance": float(np.mean([ref64[i][3] for i in range(n)])),
            "note": "GPU results of the first instances of the timed batch in each f32 arithmetic of the library (each bit-identical to its CPU oracle) against the float64 build of the same "
                    "oracle (libm activations, no operand splitting): ONE gradient at a perturbed control sequence (rms / max error relative to the float64 gradient's largest entry) and the "
                    "FULL cold-start solve (instances with every control within abs + rel 1e-4 of the float64 solve; median / worst max|du|). The reference's own path is a further f32 "
                    "rounding (JAX on CPU, sde_control.py:6), not available here: the claim these figures support is 'no further from float64 than a plain f32 evaluation', per arithmetic. "
                    "tests/test_arithmetic_referee_cpu.py holds the same table for 64 C1-sized and 8 C2-sized instances (profiles/r5_referee.json)"}


if __name__ == "__main__":
    main()
