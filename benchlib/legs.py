"""One workload on one GPU (solver handle + device-resident inputs and outputs), the single-solve latency loop, BASELINE config 4 and the
secondary configurations of bench.py's default line."""
import os
import time

import numpy as np

from .counts import F16_MFMA_PEAK_TF, F32_MFMA_PEAK_TF, HBM_PEAK_GBS, f16_contraction_flops, roofline_of
from .verify import sample_indices


class Leg:
    """One workload on this rank's GPU: solver + device-resident inputs and outputs."""

    def __init__(self, cfg, blob, B, dev_ord, rank=0, world=1, pos=False, plant=()):
        """plant: (index, x0[13], xref[H+1][13], key[2]) tuples — instances of the batch replaced by given problems (the instance a committed
        full-length oracle result exists for: tests/golden/make_c5_fullsize.py), solved by the timed launches like every other"""
        import torch
        from sde4mbrl_px4_amd import prng
        from sde4mbrl_px4_amd import workload as W
        from sde4mbrl_px4_amd.solver import SdeMpcSolver
        self.cfg, self.blob, self.B = cfg, blob, B
        H, P, m = cfg.horizon, cfg.num_particles, cfg.num_motors
        dev = torch.device("cuda", dev_ord)
        self.solver = SdeMpcSolver(cfg, blob, max_batch=B, device=dev_ord)
        self.x0_h = W.random_initial_states(B, rank * B)
        self.xref_h = np.stack([W.constant_reference(W.HOVER, H) if pos else W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(B)])
        # noise: drawn on the device from per-instance threefry keys (SPEC.md 7; the m_mpc path), launch seed 10 (iris_sdectrl.launch:8)
        self.keys = prng.split(prng.PRNGKey(10), world * B)[rank * B:(rank + 1) * B].copy()
        for i, px0, pxref, pkey in plant:
            self.x0_h[i], self.xref_h[i], self.keys[i] = px0, pxref, pkey
        yk, info0 = self.solver.reset()
        self.u0_h = np.tile(yk[None], (B, 1, 1))
        self.s0 = float(info0["stepsize"])
        self.stream = torch.cuda.current_stream().cuda_stream
        self.x0 = torch.from_numpy(self.x0_h).to(dev)
        self.xref = torch.from_numpy(self.xref_h).to(dev)
        self.noise = torch.empty(self.solver.lib.sdempc_noise_dev_floats(self.solver._h, B), dtype=torch.float32, device=dev)
        self.solver.noise_from_keys_dev(self.keys, self.noise.data_ptr(), self.stream)
        self.u0 = torch.from_numpy(self.u0_h).to(dev)
        self.step_in = torch.full((B,), self.s0, dtype=torch.float32, device=dev)
        self.uopt = torch.empty((B, H, m), dtype=torch.float32, device=dev)
        self.xevol = torch.empty((B, H + 1, 13), dtype=torch.float32, device=dev)
        self.info = torch.empty((B, 8), dtype=torch.float32, device=dev)

    def step(self, solver=None, out=None):
        u, x, i = out or (self.uopt, self.xevol, self.info)
        (solver or self.solver).solve_dev(self.B, self.x0.data_ptr(), self.xref.data_ptr(), self.noise.data_ptr(), self.u0.data_ptr(),
                                          self.step_in.data_ptr(), u.data_ptr(), x.data_ptr(), i.data_ptr(), self.stream)

    def solve_one(self, solver, i):
        """instance i of the batch alone (B = 1 launch of the same C-ABI entry point), outputs into the batch's own rows"""
        nv = self.noise.view(self.B, -1)
        solver.solve_dev(1, self.x0[i:].data_ptr(), self.xref[i:].data_ptr(), nv[i:].data_ptr(), self.u0[i:].data_ptr(), self.step_in[i:].data_ptr(),
                         self.uopt[i:].data_ptr(), self.xevol[i:].data_ptr(), self.info[i:].data_ptr(), self.stream)

    def timed_events(self, reps):
        """reps launches timed one by one with HIP events on the launch stream; returns (ms list, work counters per solve)"""
        import torch
        torch.cuda.synchronize()
        self.solver.work_counters(reset=True)
        ms = []
        for _ in range(reps):
            self.step()
            ms.append(self.solver.last_kernel_ms())
        torch.cuda.synchronize()
        self.solver.solve_status()
        w_solves, w_grads, w_fwd = self.solver.work_counters()
        if w_solves != self.B * reps:
            raise SystemExit(f"bench.py: {w_solves} solves counted on the device for {reps} launches of {self.B} instances")
        return ms, w_grads / max(w_solves, 1), w_fwd / max(w_solves, 1)

    def host_outputs(self):
        return self.uopt.cpu().numpy(), self.xevol.cpu().numpy(), self.info.cpu().numpy()

    def slots(self):
        return 6 * self.solver.get_option("device_cus")

    def close(self):
        self.solver.close()


def latency_of(L, solver, reps, warm):
    """p50 / p95 protocol of SURVEY.md §8(d): one problem instance after the other, host timestamps around a device sync"""
    import torch
    lat = []
    for r in range(-warm if reps > 0 else 0, reps):
        i = r % L.B
        torch.cuda.synchronize()
        t = time.perf_counter()
        L.solve_one(solver, i)
        torch.cuda.synchronize()
        if r >= 0:
            lat.append((time.perf_counter() - t) * 1e3)
        solver.solve_status()     # raises if a grid barrier of the cooperative layout gave up (results would be invalid)
    return lat, (solver.last_kernel_name() if lat else None), solver.layout_fallbacks()


def config4_leg(L, cfg, blob, dev, dev_ord, world, reps, mlp_dtype, sync_all, force_dist):
    """BASELINE config 4 (N > 1 only): ONE instance per GPU, all ranks solving theirs at the same time (barrier-aligned ticks, duration of a
    tick = the slowest rank's); f32 latency layouts like the single-GPU p50 (a single instance is a latency problem)"""
    import torch
    from sde4mbrl_px4_amd.solver import SdempcError
    from sde4mbrl_px4_amd.dist import max_over_ranks_each
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    s1 = L.solver if mlp_dtype == "f32" else SdeMpcSolver(cfg.replace(mlp_dtype="f32"), blob, max_batch=8, device=dev_ord)
    durs, gave_up, last_ok = [], 0, None
    for r in range(-3, reps):
        i = (r + 3) % L.B
        sync_all()
        t = time.perf_counter()
        L.solve_one(s1, i)
        torch.cuda.synchronize()
        d = (time.perf_counter() - t) * 1e3
        ok = True
        try:
            s1.solve_status()
        except SdempcError as e:
            # a grid barrier gave up (the GPU is shared with another rank: test boxes only): the handle continues in the tile layout and the tick is
            # not a measurement; anything else (a ticket mismatch, a launch failure) is an error of the run
            if "barrier" not in str(e):
                raise
            gave_up += 1
            ok = False
        if r >= 0:
            durs.append(d if ok else float("inf"))
        if ok:
            last_ok = i
    kernel, fallbacks = s1.last_kernel_name(), s1.layout_fallbacks()
    # what this rank's last good tick produced (rows of the batch's output arrays, in the f32 arithmetic of the latency layouts): every rank
    # hands it to its own checker (bench.py: verified_by_rank)
    check = None
    if last_ok is not None:
        check = {"idx": last_ok, "cfg": cfg.replace(mlp_dtype="f32"),
                 "got": (L.uopt[last_ok].cpu().numpy(), L.xevol[last_ok].cpu().numpy(), L.info[last_ok].cpu().numpy())}
    if s1 is not L.solver:
        s1.close()
    # a tick counts only if every rank measured it: +inf marks a dropped one and survives the MAX over the ranks
    ticks = [t for t in max_over_ranks_each(durs, device=dev, force=force_dist) if np.isfinite(t)]
    if not ticks:
        return {"instances": world, "ticks": 0, "value": None, "barrier_give_ups_rank0": gave_up, "kernel": kernel,
                "note": "no tick completed on every rank without a barrier give-up"}, check
    return {"instances": world, "ticks": len(ticks), "ticks_dropped": len(durs) - len(ticks), "p50_tick_ms": float(np.median(ticks)), "p95_tick_ms": float(np.percentile(ticks, 95)),
            "value": world / (float(np.median(ticks)) * 1e-3), "unit": "solves/s", "kernel": kernel, "layout_fallbacks_rank0": fallbacks, "barrier_give_ups_rank0": gave_up, "mlp_dtype": "f32",
            "note": "BASELINE config 4: one Iris H=50 P=128 instance per GPU (random initial states), every rank solving its own at the same time; a tick "
                    "lasts as long as its slowest rank; no data-path collective (the weights were broadcast once at start)"}, check


def committed_instance(root, config, mlp, math):
    """(plant tuple for index -1, golden arrays) of the full-length oracle result committed for this configuration and arithmetic, or None"""
    import importlib.util
    gdir = os.path.join(root, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_c5_fullsize", os.path.join(gdir, "make_c5_fullsize.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    if (config, mlp, math) not in mk.COMMITTED or not os.path.exists(mk.golden_path(mlp, math, config)):
        return None
    cfg, x0, xref, key = mk.problem(mlp, math, config)
    return (x0[0], xref[0], key[0]), dict(np.load(mk.golden_path(mlp, math, config))), os.path.basename(mk.golden_path(mlp, math, config))


def other_config_legs(root, main_mlp, math_mode, B, dev, dev_ord, cfg_of, V, verify, progress):
    """C2 in the f32 chain, C3, C5 (main arithmetic and f16): one or two timed launches each, roofline fractions, and a check of the timed
    full-length launch itself: against the committed full-length oracle result where one exists for the configuration and arithmetic (the instance
    is planted into the batch: C5, C3), by the live checker otherwise (a sample handed to the verifier)"""
    import torch
    from sde4mbrl_px4_amd import synthetic_hexa, synthetic_iris
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    cdir = os.path.join(root, "configs")
    legs = [("c2_f32_chain", "c2", os.path.join(cdir, "c2_iris_traj_h50_p128.yaml"), "f32", B, 2),
            ("c3", "c3", os.path.join(cdir, "c3_hexa_traj_h50_p256.yaml"), main_mlp, 6144, 2),
            ("c5_f32x3" if main_mlp == "f32x3" else "c5_f32", "c5", os.path.join(cdir, "c5_iris_traj_h200_p1024.yaml"), main_mlp, 768, 1),
            ("c5_f16", "c5", os.path.join(cdir, "c5_iris_traj_h200_p1024.yaml"), "f16", 768, 1)]
    if main_mlp == "f32":
        legs = legs[1:]
    iris_blob, hexa_blob = synthetic_iris().to_blob(), synthetic_hexa().to_blob()
    others = {}
    for name, config, path, mlp, Bl, reps in legs:
        progress(f"other configuration {name}: {Bl} instances, {reps} timed launch(es)")
        c2 = cfg_of(path, mlp)
        bl = iris_blob if c2.num_motors == 4 else hexa_blob
        com = committed_instance(root, config, mlp, math_mode) if verify != 0 else None
        Lg = Leg(c2, bl, Bl, dev_ord, plant=[(Bl - 1,) + com[0]] if com else ())
        Lg.step(); torch.cuda.synchronize()                       # warm-up launch
        ms, ng, nf = Lg.timed_events(reps)
        kn = Lg.solver.last_kernel_name()
        km = float(np.mean(ms))
        tf, gbs, _, _, _ = roofline_of(c2, Bl, km, ng, nf)
        uo, xo, io = Lg.host_outputs()
        rec = {"config": os.path.basename(path), "mlp_dtype": mlp, "math_mode": math_mode, "instances": Bl, "launches_timed": reps, "value": Bl / (km * 1e-3), "unit": "solves/s",
               "kernel_ms": km, "kernel": kn, "roofline_frac": tf / F32_MFMA_PEAK_TF, "roofline_hbm_frac": gbs / HBM_PEAK_GBS,
               "N_it_mean": float(io[:, 2].mean()), "N_grad_evaluated_mean": ng, "N_forward_rollouts_mean": nf}
        if name == "c3":
            # a lone C3 instance (P = 256, 300 controls) in the f32 latency layouts, as the headline's p50 is for C2: four groups of 64 workgroups
            s1 = SdeMpcSolver(cfg_of(path, "f32"), bl, max_batch=1, device=dev_ord)
            lat, kn1, _ = latency_of(Lg, s1, 40, 4)
            rec.update({"p50_solve_latency_ms": float(np.median(lat)), "p95_solve_latency_ms": float(np.percentile(lat, 95)), "latency_kernel": kn1, "latency_mlp_dtype": "f32"})
            s1.close()
        if mlp == "f16":
            f16_tf = f16_contraction_flops(c2, ng, nf) * Bl / (km * 1e-3) / 1e12
            rec["matrix_pipe_use_f16"] = {"achieved": f16_tf, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": f16_tf / F16_MFMA_PEAK_TF,
                                          "note": "NOT a roofline of this kernel: K = 6 and K = 32 contractions of a 32-wide MLP cannot fill the matrix pipe (under 2 % of the "
                                                  "2.5 PFLOP/s dense f16 peak by construction); the kernel is bound by the f32 vector work beside them (roofline_frac)"}
        if verify != 0:
            if com:
                # the last instance of the timed batch IS the problem whose full-length oracle result is committed: the timed launch's own output, bit for bit
                V.add_committed(name, Bl - 1, (uo[Bl - 1], xo[Bl - 1], io[Bl - 1]), com[1], com[2])
                rec["verified_how"] = (f"full-length, bit for bit: instance {Bl - 1} of the timed launch is the problem tests/golden/{com[2]} holds the CPU oracle's "
                                       f"full-length result for (N_it {com[1]['info'][2]:.0f}, N_ls {com[1]['info'][7]:.0f}; tests/golden/make_c5_fullsize.py)")
            elif name.startswith("c5"):
                # no committed result for this arithmetic: the SAME instances are solved once more with three iterations from a step size at which
                # all three take steps and that launch is checked by the live oracle
                c3it = c2.replace(max_iter=3, max_no_improvement_iter=3)
                s3 = SdeMpcSolver(c3it, bl, max_batch=Bl, device=dev_ord)
                u3, x3, i3 = torch.empty_like(Lg.uopt), torch.empty_like(Lg.xevol), torch.empty_like(Lg.info)
                st3 = torch.full((Bl,), 1e-11, dtype=torch.float32, device=dev)
                s3.solve_dev(Bl, Lg.x0.data_ptr(), Lg.xref.data_ptr(), Lg.noise.data_ptr(), Lg.u0.data_ptr(), st3.data_ptr(), u3.data_ptr(), x3.data_ptr(), i3.data_ptr(), Lg.stream)
                torch.cuda.synchronize()
                V.add(name, c3it, bl, [Bl - 1], Lg.x0_h, Lg.xref_h, Lg.keys, Lg.u0_h, 1e-11, (u3.cpu().numpy(), x3.cpu().numpy(), i3.cpu().numpy()))
                rec["verified_how"] = "3-iteration launch of the same instances (same kernel instantiation, step size 1e-11), last instance, bit for bit by the live oracle"
                s3.close()
            else:
                vi = sample_indices(Bl, Lg.slots(), n_initial=1, n_drawn=1)
                V.add(name, c2, bl, vi, Lg.x0_h, Lg.xref_h, Lg.keys, Lg.u0_h, Lg.s0, (uo, xo, io))
                rec["verified_how"] = "the timed full-length launch, bit for bit by the live oracle"
        others[name] = rec
        Lg.close()
    return others


def other_math_mode_legs(L, cfg, blob, dev_ord, uopt_h):
    """The OTHER math mode on the same instances (a warm-up and a timed launch in the run's contraction arithmetic; a short single-solve latency
    loop in the f32 contractions, where the latency layouts exist): solves/s, p50, and how far its controls are from this run's (north star: 1e-4).
    Both modes are bit-identical to the oracle in their own arithmetic (SPEC.md 3 / 10 + 10a); this leg shows what the choice costs and changes."""
    import torch
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    B = L.B
    other = "exact" if cfg.math_mode == "fast" else "fast"
    modes, disputed = {}, None
    u2 = torch.empty_like(L.uopt); x2 = torch.empty_like(L.xevol); i2 = torch.empty_like(L.info)
    keep = (L.uopt.clone(), L.xevol.clone(), L.info.clone())
    for mlp in dict.fromkeys((cfg.mlp_dtype, "f32")):
        s2 = SdeMpcSolver(cfg.replace(math_mode=other, mlp_dtype=mlp), blob, max_batch=B, device=dev_ord)
        for _ in range(2):                  # (a first launch of these kernels measured 7 % slow)
            L.step(s2, (u2, x2, i2))
            ms2 = s2.last_kernel_ms()
        torch.cuda.synchronize()
        rec = {"value": B / (ms2 * 1e-3), "unit": "solves/s", "kernel": s2.last_kernel_name()}
        if mlp == cfg.mlp_dtype:
            du = np.abs(u2.cpu().numpy() - uopt_h).reshape(B, -1)
            ok = np.all(du <= 1e-4 + 1e-4 * np.abs(uopt_h).reshape(B, -1), axis=1)
            rec.update({"max_abs_du_vs_this_run_median": float(np.median(du.max(axis=1))), "max_abs_du_vs_this_run_worst": float(du.max()),
                        "instances_within_1e-4_of_this_run": float(ok.mean())})
            # the instances on which the two math modes disagree most: bench.py hands them to the float64 referee (which mode is closer?)
            worst = np.argsort(-du.max(axis=1))[:8]
            disputed = {"idx": [int(i) for i in worst], "u_other": u2.cpu().numpy()[worst], "other": f"{mlp}/{other}"}
        if mlp == "f32":
            lat, kn, _ = latency_of(L, s2, 60, 5)
            rec.update({"p50_ms": float(np.median(lat)), "latency_kernel": kn})
        modes[f"{mlp}/{other}"] = rec
        s2.close()
    L.uopt.copy_(keep[0]); L.xevol.copy_(keep[1]); L.info.copy_(keep[2])
    return dict(modes, note="same instances, cold-start 200-iteration solves in the other math mode; controls against this run's "
                            "(abs + rel 1e-4, the north star's tolerance: over 200 iterations a different rounding flips a line-search decision in a few per cent of the instances; "
                            "disputed_instances_vs_float64 has the float64 referee's view of the instances that differ most)"), disputed


def float64_referee_leg(L, cfg, blob, dev_ord, n, ug):
    """The first n instances of the timed batch in each of the four f32 arithmetics on the GPU — the full cold-start solve and ONE gradient at the
    perturbed control sequences ug — for bench.py's vs_float64 table (benchlib/referee.py). Returns {arithmetic: (grad [n,H,m], cost [n], uopt [n,H,m])}."""
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    from .referee import ARITHMETICS, name
    s0 = np.full(n, L.s0, np.float32)
    out = {}
    for mlp, mm in ARITHMETICS:
        S = SdeMpcSolver(cfg.replace(mlp_dtype=mlp, math_mode=mm), blob, max_batch=n, device=dev_ord)
        noise = S.noise_from_keys(L.keys[:n])                # the canonical tensors the device draws from the keys (bit-identical to the oracle's: test_gpu_parity)
        c, g = S.grad(L.x0_h[:n], ug, L.xref_h[:n], noise)
        u = S.solve_keys(L.x0_h[:n], L.xref_h[:n], L.keys[:n], L.u0_h[:n], s0)[0]
        out[name(mlp, mm)] = (np.asarray(g, np.float64), np.asarray(c, np.float64), u)
        S.close()
    return out
