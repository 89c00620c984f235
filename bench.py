#!/usr/bin/env python3
"""Benchmark of the MPC inner loop on MI355X: MPC solves/sec (+ p50 solve latency), Iris H=50 P=128.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched under torch.distributed.run,
one rank per GPU over RCCL). A "step" = one launch of the solve kernel over one batch of B independent
MPC problem instances per GPU (synthetic initial states; inputs already resident in HBM). Weak scaling:
B per GPU is fixed, instances are sharded one batch per GPU with no data-path collective; rank 0
broadcasts the shared model blob once over RCCL at start-up. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
F32_MFMA_PEAK_TF = 157.3     # dense f32-input MFMA peak (= f32 vector peak), same guide


def algorithmic_counts(cfg, n_it, n_ls):
    """SURVEY.md §8(d) per-solve algorithmic bytes and flops (n_w = 6 noisy dims, f32)."""
    P, H, m = cfg.num_particles, cfg.horizon, cfg.num_motors
    nw = 6
    b_grad = 4 * (P * H * nw + 2 * P * (H + 1) * 13 + P * H * nw + 2 * H * m + (H + 1) * 13)
    b_ls = 4 * (P * H * nw + H * m + (H + 1) * 13)
    w_bytes = 4 * 2120
    # init-cost rollout and final mean-trajectory rollout are forward-only passes too
    bytes_solve = n_it * b_grad + (n_ls + 2) * b_ls + w_bytes
    f_step = 2 * ((6 + m) * 32 + 32 * 32 + 32 * 6) + 2 * (6 * 32 + 32 * 1)   # drift + density nets, forward
    flops_solve = f_step * P * H * (2 * n_it + n_ls + 2)
    return bytes_solve, flops_solve, b_grad, b_ls


def checkpoint_bytes(cfg, n_it):
    """Implementation stream on top of the algorithmic bytes: the gradient's forward sweep checkpoints the
    layer-2 activations + 5 step scalars per particle-step (1280 floats per 32-particle group and step),
    written once and read once per gradient evaluation."""
    G = (cfg.num_particles + 31) // 32
    return int(n_it * 2 * G * cfg.horizon * 1280 * 4)


def effective_cores():
    """Host cores this process may actually use: min(os.cpu_count, affinity, cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, model, n_threads, x0, xref, noise, u0, s0):
    """Oracle (CPU restatement, kind 'port') on the host cores: one solve per thread, wall clock."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oracles = [orc.Oracle(cfg, model) for _ in range(n_threads)]
    out = [None] * n_threads

    reps = 2   # solves per thread: keeps the sample at ~10-15 s of wall time

    def work(i):
        for _ in range(reps):
            out[i] = oracles[i].solve(x0[i], xref[i], noise[i], u0[i], s0)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    return reps * n_threads / dt, dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=3072, help="MPC problem instances per GPU per step (3 workgroups per CU x 256 CUs x 4 rounds)")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--latency-reps", type=int, default=1000)
    ap.add_argument("--latency-warmup", type=int, default=20)
    ap.add_argument("--mlp-dtype", default="f32", choices=["f32", "f16"],
                    help="f32: bit-reproducible path (default, the reported metric); f16: fp16-operand MLP contractions (SPEC.md 9)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sde4mbrl_px4_amd import load_mpc_config, synthetic_hexa, synthetic_iris
    from sde4mbrl_px4_amd import workload as W
    from sde4mbrl_px4_amd.solver import SdeMpcSolver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # test hooks (single-GPU boxes): SDEMPC_BENCH_DEVICE pins every rank to one ordinal, SDEMPC_BENCH_BACKEND=gloo
    # replaces RCCL; the driver's multi-GPU runs use neither (one GPU per rank, backend nccl = RCCL over xGMI)
    dev_ord = int(os.environ.get("SDEMPC_BENCH_DEVICE", local_rank))
    backend = os.environ.get("SDEMPC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_ord)
    dev = torch.device("cuda", dev_ord)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = load_mpc_config(args.config).replace(mlp_dtype=args.mlp_dtype)
    H, P, m, B = cfg.horizon, cfg.num_particles, cfg.num_motors, args.batch
    # shared model: rank 0 builds it, RCCL broadcast over xGMI (read-only weights are the only shared data)
    from sde4mbrl_px4_amd.dist import broadcast_blob, max_over_ranks
    blob = (synthetic_iris() if m == 4 else synthetic_hexa()).to_blob() if rank == 0 else b""
    blob = broadcast_blob(blob, src=0, device=dev)

    solver = SdeMpcSolver(cfg, blob, max_batch=B, device=dev_ord)
    # synthetic inputs, distinct per rank, resident in HBM before timing
    seed0 = rank * B
    x0_h = W.random_initial_states(B, seed0)
    xref_h = np.stack([W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(B)])
    # noise: drawn on the device from per-instance threefry keys (SPEC.md §7; the m_mpc path), launch seed 10
    # (iris_sdectrl.launch:8) split into one key per instance of the whole job
    from sde4mbrl_px4_amd import prng
    keys = prng.split(prng.PRNGKey(10), world * B)[rank * B:(rank + 1) * B]
    nd = solver.lib.sdempc_noise_dev_floats(solver._h, B)
    yk, info0 = solver.reset()
    u0_h = np.tile(yk[None], (B, 1, 1))
    s0 = float(info0["stepsize"])
    x0 = torch.from_numpy(x0_h).to(dev)
    xref = torch.from_numpy(xref_h).to(dev)
    noise = torch.empty(nd, dtype=torch.float32, device=dev)
    solver.noise_from_keys_dev(keys, noise.data_ptr(), torch.cuda.current_stream().cuda_stream)
    u0 = torch.from_numpy(u0_h).to(dev)
    step_in = torch.full((B,), s0, dtype=torch.float32, device=dev)
    uopt = torch.empty((B, H, m), dtype=torch.float32, device=dev)
    xevol = torch.empty((B, H + 1, 13), dtype=torch.float32, device=dev)
    info = torch.empty((B, 8), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        solver.solve_dev(B, x0.data_ptr(), xref.data_ptr(), noise.data_ptr(), u0.data_ptr(), step_in.data_ptr(),
                         uopt.data_ptr(), xevol.data_ptr(), info.data_ptr(), stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    elapsed = max_over_ranks(elapsed, device=dev)
    # per-launch kernel duration from HIP events, measured live on the launch stream (separate launches)
    ev_ms = []
    for _ in range(min(args.steps, 3)):
        step()
        ev_ms.append(solver.last_kernel_ms())
    torch.cuda.synchronize()
    info_h = info.cpu().numpy()
    n_it = float(info_h[:, 2].mean())
    n_ls = float(info_h[:, 7].mean())

    if rank == 0:
        solves = world * B * args.steps
        value = solves / elapsed
        bytes_solve, flops_solve, b_grad, b_ls = algorithmic_counts(cfg, n_it, n_ls)
        k_ms = float(np.mean(ev_ms))
        ach_gbs = bytes_solve * B / (k_ms * 1e-3) / 1e9
        ach_tf = flops_solve * B / (k_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                key = f"{os.path.basename(args.config)}:B{B}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # p50 / p95 latency of a single solve (B=1 launches), outside the timed region: SURVEY.md §8(d) protocol, >= 20 warm-up
        # and >= 1000 timed solves by default, one problem instance after the other (host timestamps around a device sync)
        lat = []
        nv = noise.view(B, -1)
        for r in range(-args.latency_warmup, args.latency_reps):
            i = r % B
            torch.cuda.synchronize()
            t = time.perf_counter()
            solver.solve_dev(1, x0[i:].data_ptr(), xref[i:].data_ptr(), nv[i:].data_ptr(), u0[i:].data_ptr(), step_in[i:].data_ptr(),
                             uopt[i:].data_ptr(), xevol[i:].data_ptr(), info[i:].data_ptr(), stream)
            torch.cuda.synchronize()
            if r >= 0:
                lat.append((time.perf_counter() - t) * 1e3)
        out = {
            "metric": "MPC solves/sec, Iris H=50 P=128 (p50 solve latency in p50_solve_latency_ms)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.mlp_dtype == "f32" else "f16 MLP operands / f32 accumulate and state", "data": "synthetic",
            "config": {"workload": f"{os.path.basename(args.config)}: H={H} P={P} m={m}, {B} independent MPC instances per GPU per step, "
                                   f"cold-start solves from the hover guess, max_iter={cfg.max_iter} maxls={cfg.ls_maxls}",
                       "instances_per_gpu": B, "noise": "threefry2x32 keys (seed 10 split per instance), normal draws generated on the device", "N_it_mean": n_it, "N_ls_mean": n_ls, "parallelism": f"instances sharded over {world} GPU(s), no data-path collective"},
            "p50_solve_latency_ms": float(np.median(lat)),
            "p95_solve_latency_ms": float(np.percentile(lat, 95)),
            "latency_reps": len(lat),
            "p50_solve_latency_note": "one instance alone on the GPU (B = 1 launch of the same C-ABI entry point): the library spreads it over ceil(P/4) x 7 workgroups "
                                      "(one particle per wave; three line-search trials and the candidate gradients of the next iteration evaluated at once); bit-identical results",
            "p50_batch_latency_ms": float(np.median(ev_ms)),
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach_tf / F32_MFMA_PEAK_TF,
                         "traffic": traffic, "kernel": "sdempc::exact::sdempc_solve_kernel<sdempc::exact::%s, %d, %s, false, 0, false>" % ("TeamBlock" if P > 32 else "TeamWave", m if m in (4, 6) else 8, "true" if args.mlp_dtype == "f16" else "false"), "kernel_ms": k_ms,
                         "note": "f32-exact path: MLP contractions on v_mfma_f32_32x32x2_f32 (157.3 TF dense peak = f32 vector peak); "
                                 "algorithmic flops = SURVEY §8d MLP formula x P*H*(2*N_it+N_ls+2)"},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                             "traffic": traffic, "bytes_per_solve": bytes_solve, "B_grad": b_grad, "B_ls": b_ls,
                             "checkpoint_bytes_per_solve": checkpoint_bytes(cfg, n_it),
                             "note": "achieved = SURVEY 8d algorithmic bytes / kernel time; measured traffic additionally contains the activation-checkpoint stream"},
        }
        if not args.no_cpu_baseline and world == 1:
            nthr = args.cpu_threads or min(effective_cores(), 64)
            from sde4mbrl_px4_amd.model import RotorSDEModel  # noqa: F401
            noise_h = W.make_noise(nthr, P, H, 777)
            xb = W.random_initial_states(nthr, 0)
            xr = np.stack([W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(nthr)])
            ub = np.tile(yk[None], (nthr, 1, 1))
            v, dt, _ = cpu_baseline(cfg, blob, nthr, xb, xr, noise_h, ub, s0)
            out["cpu_baseline"] = {"value": v, "unit": "solves/s", "cores": nthr, "kind": "port",
                                   "sample": f"{2 * nthr} solves of the same workload (two per thread, one thread per usable host core: os.cpu_count {os.cpu_count()}, "
                                             f"cgroup/affinity limit {effective_cores()}; {dt:.1f} s wall) by the C oracle "
                                             "(CPU restatement, not the reference JAX path: that cannot run here)"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    solver.close()


if __name__ == "__main__":
    main()
