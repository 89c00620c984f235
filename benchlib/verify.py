"""bench.py's CPU legs: the oracle (oracle/, test infrastructure) as the checker of timed launches and as the reported cpu_baseline. Nothing here runs
inside a timed region, and nothing under sde4mbrl_px4_amd/ imports it."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def cpu_oracle():
    """The CPU restatement (oracle/, test infrastructure): bench.py touches it only in its CPU legs — as the checker of the
    timed launch's outputs and as the reported cpu_baseline, never inside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    return orc


def words_differ(a, b):
    """f32 words whose bits differ (NaNs compared as a class: x86 and gfx950 produce different NaN signs, SPEC.md §3.7)."""
    fa, fb = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    both_nan = np.isnan(fa) & np.isnan(fb)
    return int(((fa.view(np.uint32) != fb.view(np.uint32)) & ~both_nan).sum())


def sample_indices(B, slots, n_initial=4, n_drawn=6):
    """Instances of a launch to verify: some of the teams' initial assignments, some that a persistent launch hands out by ticket
    (b >= slots; evenly spread), and the last one."""
    idx = list(range(min(n_initial, B)))
    if B > slots + 1:
        idx += [int(v) for v in np.linspace(slots, B - 2, n_drawn)]
    if B - 1 not in idx:
        idx.append(B - 1)
    return sorted(set(idx))


def cpu_solve_instances(cfg, model, n_threads, x0, xref, keys, u0, s0, fast=False, native=False):
    """CPU leg, kind 'port': the C oracle (CPU restatement of SPEC.md) solves the given instances — instances of the GPU batch, noise
    derived from the same threefry keys — one solve at a time per thread. fast=False: the bit-exact checker (what the GPU results are
    compared with; native=True takes its -O3 -march=native build, same source and same bits); fast=True: the same source built as the
    particle-vectorised timing build (oracle/Makefile: liborc_vec.so, 16 particles per call through GCC vector extensions, contraction
    allowed: tolerance parity), the credible CPU timing. Returns (solves/s, wall s, outputs)."""
    orc = cpu_oracle()
    n = len(x0)
    O = [orc.Oracle(cfg, model, vec=fast, fast=native and not fast) for _ in range(n_threads)]
    P, H = cfg.num_particles, cfg.horizon
    out = [None] * n
    nxt = [0]
    lock = threading.Lock()

    def work(i):
        while True:
            with lock:
                j = nxt[0]; nxt[0] += 1
            if j >= n:
                return
            noise = orc.noise_from_key(keys[j], P, H)
            out[j] = O[i].solve(x0[j], xref[j], noise, u0[j], s0)[:3]

    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    return n / dt, dt, out


class Verifier:
    """Background CPU work of bench.py: worker threads solve sampled instances with the CPU oracle (ctypes releases the GIL) while the main thread
    goes on with the GPU legs; results are collected at the end. Three kinds of jobs share the pool, served in the order they were queued:
      * checks   (add):          an instance of a timed launch solved by the bit-exact checker and compared word for word with the GPU's outputs;
      * referee  (add_referee):  the float64 build of the oracle on instances of the timed batch (bench.py's vs_float64 field);
      * baseline (add_baseline): instances solved by the particle-vectorised timing build — the reported cpu_baseline. EVERY worker thread solves its own
        quota of them once the shared queue has nothing left for it: threads whose checks end early start early, the rate is the sum over the threads of
        (solves of that thread / the time it spent on them), every core being busy throughout (with checks or with baseline solves) except in the tail.
    A worker that dies (an exception inside the oracle or the comparison) is recorded: bench.py exits non-zero when any leg has fewer checked
    instances than it asked for."""

    def __init__(self, n_threads):
        self.jobs, self.results, self.lock = [], {}, threading.Lock()
        self.n_threads, self.threads, self.t0 = max(1, n_threads), [], None
        self.errors = []
        self.referee, self.baseline = {}, {"solves": 0, "busy": {}, "outs": {}}
        self.baseline_spec = None

    def add(self, leg, cfg, blob, idx, x0, xref, keys, u0, s0, got):
        """got: (uopt, xevol, info) host arrays of the WHOLE batch; idx: instances to check"""
        with self.lock:
            self.results.setdefault(leg, {"idx": [int(i) for i in idx], "bad_words": 0, "done": 0, "cpu_s": 0.0})
            for i in idx:
                self.jobs.append(("check", leg, cfg, blob, int(i), x0[i], xref[i], keys[i], u0[i], s0, got[0][i].copy(), got[1][i].copy(), got[2][i].copy()))
            if self.threads:
                self.cv.notify_all()

    def add_committed(self, leg, i, got, golden, what):
        """an instance whose full-length oracle result is a committed fixture (tests/golden/*_fullsize_*.npz): compared at once, no CPU time"""
        bad = words_differ(got[0], golden["uopt"]) + words_differ(got[1], golden["xevol"]) + words_differ(got[2], golden["info"])
        with self.lock:
            self.results[leg] = {"idx": [int(i)], "bad_words": bad, "done": 1, "cpu_s": 0.0, "committed": what}

    def add_referee(self, cfg, blob, idx, x0, xref, keys, u0, ug, s0):
        """float64 solve (from u0) and, when ug is given, float64 gradient (at ug[i]) of instances idx"""
        with self.lock:
            for i in idx:
                self.jobs.append(("f64", "referee", cfg, blob, int(i), x0[i], xref[i], keys[i], u0[i], s0, None if ug is None else ug[i], None, None))
            if self.threads:
                self.cv.notify_all()

    def add_baseline(self, cfg, blob, per_thread, x0, xref, keys, u0, s0):
        """per_thread solves by every worker thread (thread t: instances t * per_thread ... of the arrays given), behind whatever the shared queue holds"""
        with self.lock:
            self.baseline_spec = (cfg, blob, int(per_thread), x0, xref, keys, u0, s0)

    def start(self):
        """start the workers; jobs added later are picked up too, until join()"""
        orc = cpu_oracle()
        self.t0 = time.time()
        self.cv, self.closed, self.nxt = threading.Condition(self.lock), False, 0
        oracles = {}

        def oracle_of(tid, kind, leg, cfg, blob):
            with self.lock:
                O = oracles.get((tid, kind, leg))
            if O is None:
                if kind == "f64":
                    O = orc.Oracle(cfg.replace(mlp_dtype="f32", math_mode="exact"), blob, double=True)
                elif kind == "vec":
                    O = orc.Oracle(cfg, blob, vec=True)
                else:
                    O = orc.Oracle(cfg, blob)                  # the bit-exact checker build the parity tests use
                with self.lock:
                    oracles[(tid, kind, leg)] = O
            return O

        def work(tid):
            while True:
                with self.cv:
                    while self.nxt >= len(self.jobs) and not self.closed:
                        self.cv.wait()
                    if self.nxt >= len(self.jobs):
                        break
                    job = self.jobs[self.nxt]; self.nxt += 1
                kind, leg, cfg, blob, i, x0, xref, key, u0, s0, gu, gx, gi = job
                try:
                    O = oracle_of(tid, kind, leg, cfg, blob)
                    t = time.time()
                    noise = orc.noise_from_key(key, cfg.num_particles, cfg.horizon)
                    if kind == "check":
                        uo, xe, io = O.solve(x0, xref, noise, u0, s0)[:3]
                        bad = words_differ(gu, uo) + words_differ(gx, xe) + words_differ(gi, io)
                        with self.lock:
                            r = self.results[leg]; r["bad_words"] += bad; r["done"] += 1; r["cpu_s"] += time.time() - t
                    elif kind == "f64":
                        c64, g64 = O.grad(x0, gu, xref, noise) if gu is not None else (None, None)
                        u64 = O.solve(x0, xref, noise, u0, s0)[0]
                        with self.lock:
                            if gu is not None or i not in self.referee:       # (an instance may be asked for twice: with its gradient and as a disputed one)
                                self.referee[i] = (g64, c64, u64, time.time() - t)
                except Exception as e:       # the job stays "not done": bench.py reports the leg as unverified and exits non-zero
                    with self.lock:
                        self.errors.append(f"{leg}[{i}]: {type(e).__name__}: {e}")
            spec = self.baseline_spec
            if spec is None:
                return
            cfg, blob, per_thread, x0, xref, keys, u0, s0 = spec          # this thread's own share of the cpu_baseline
            try:
                O = oracle_of(tid, "vec", "baseline", cfg, blob)
                for k in range(per_thread):
                    i = tid * per_thread + k
                    if i >= len(x0):
                        break
                    t = time.time()
                    noise = orc.noise_from_key(keys[i], cfg.num_particles, cfg.horizon)
                    out = O.solve(x0[i], xref[i], noise, u0[i], s0)[:3]
                    with self.lock:
                        b = self.baseline
                        b["solves"] += 1; b["outs"][i] = out
                        n, busy = b["busy"].get(tid, (0, 0.0)); b["busy"][tid] = (n + 1, busy + time.time() - t)
            except Exception as e:
                with self.lock:
                    self.errors.append(f"baseline[thread {tid}]: {type(e).__name__}: {e}")

        self.threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(self.n_threads)]
        [t.start() for t in self.threads]

    def join(self):
        """no more jobs: wait for the queue to drain"""
        if not self.threads:
            return 0.0
        with self.cv:
            self.closed = True
            self.cv.notify_all()
        [t.join() for t in self.threads]
        return time.time() - self.t0 if self.t0 else 0.0

    def incomplete(self):
        """legs whose checked count is short of what was asked (a worker died)"""
        return {leg: (r["done"], len(r["idx"])) for leg, r in self.results.items() if r["done"] != len(r["idx"])}

    def baseline_rate(self):
        """(solves/s summed over the threads, threads that took part, solves, longest per-thread busy time)"""
        busy = self.baseline["busy"]
        if not busy:
            return None
        return sum(n / t for n, t in busy.values() if t > 0), len(busy), self.baseline["solves"], max(t for _, t in busy.values())


def calibrate_checker_seconds(cfg, blob, x0, xref, key, u0, s0, n_grad, n_fwd):
    """Estimated CPU seconds of ONE full solve by the bit-exact checker on this host, in this arithmetic: a three-iteration solve of the same
    instance is timed and scaled by the rollouts a full solve performs (n_grad gradient evaluations = two sweeps each, n_fwd forward rollouts:
    the device's work counters)."""
    orc = cpu_oracle()
    c3 = cfg.replace(max_iter=3, max_no_improvement_iter=3)
    O = orc.Oracle(c3, blob)
    noise = orc.noise_from_key(key, cfg.num_particles, cfg.horizon)
    t = time.time()
    info = O.solve(x0, xref, noise, u0, s0)[2]
    dt = time.time() - t
    rollouts_cal = 2.0 * float(info[2]) + float(info[7]) + 2.0
    return dt * (2.0 * n_grad + n_fwd) / max(rollouts_cal, 1.0), dt


def cpu_c1_single_solve_ms(model_blob, reps=3):
    """BASELINE config 1 (Iris posctrl YAML, H=20, 32 particles, CPU path, single solve, no GPU): one thread, median wall time of a full
    cold-start solve, by the particle-vectorised timing build and by the bit-exact scalar build."""
    orc = cpu_oracle()
    from sde4mbrl_px4_amd import load_mpc_config, prng
    from sde4mbrl_px4_amd import workload as W
    cfg = load_mpc_config(os.path.join(ROOT, "configs", "c1_iris_posctrl_h20_p32.yaml"))
    x0 = W.random_initial_states(reps, 0)
    keys = prng.split(prng.PRNGKey(10), reps)
    u0 = np.tile(np.asarray(cfg.uref, np.float32)[None], (cfg.horizon, 1))
    out = {}
    for kind, O in (("vec", orc.Oracle(cfg, model_blob, vec=True)), ("scalar", orc.Oracle(cfg, model_blob))):
        ms, nit = [], []
        for r in range(reps):
            noise = orc.noise_from_key(keys[r], cfg.num_particles, cfg.horizon)
            xref = W.constant_reference(W.HOVER, cfg.horizon)
            t = time.perf_counter()
            _, _, info, _ = O.solve(x0[r], xref, noise, u0, cfg.ls_init_stepsize)
            ms.append((time.perf_counter() - t) * 1e3)
            nit.append(float(info[2]))
        out[kind] = (float(np.median(ms)), float(np.mean(nit)))
    return out, cfg


def instruction_model_check(root, rank, progress, n=1 << 20):
    """math_mode fast is verified through the oracle's RECORD of v_exp_f32 / v_rcp_f32 / v_rsq_f32 (tests/golden/transc, taken on MI355X / ROCm 7.2). On a
    GPU whose instructions answer differently (another stepping or firmware) every check of this run would fail for a reason that has nothing to do with
    the library: ask the instructions themselves first (tools/transc_study/libtransc.so, one instruction per element) and say so plainly."""
    import ctypes
    import torch
    so = os.path.join(root, "tools", "transc_study", "libtransc.so")
    if not os.path.exists(so):
        progress("instruction-model check skipped: tools/transc_study/libtransc.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
        return
    orc = cpu_oracle()
    Lp = ctypes.CDLL(so)
    Lp.transc_eval_array.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    rng = np.random.default_rng(17)
    for func, fname in ((0, "v_rcp_f32"), (1, "v_rsq_f32"), (2, "v_exp_f32")):
        if func == 2:
            x = (rng.standard_normal(n) * 6).astype(np.float32)                      # pre-activations of a step
        elif func == 0:
            x = (1.0 + np.exp2(rng.uniform(-30, 30, n))).astype(np.float32)        # 1 + 2^a
        else:
            x = rng.uniform(0.5, 2.0, n).astype(np.float32)                        # |q|^2 near 1
        xin = torch.from_numpy(x.view(np.int32)).cuda()
        o = torch.empty_like(xin)
        if Lp.transc_eval_array(func, xin.data_ptr(), x.size, o.data_ptr()) != 0:
            raise SystemExit(f"bench.py: rank {rank}: the instruction probe failed to launch")
        hw = o.cpu().numpy().view(np.uint32)
        model = orc.hw_eval(func, x).view(np.uint32)
        nbad = int((hw != model).sum())
        if nbad:
            raise SystemExit(f"bench.py: rank {rank}: the instruction model does not match this GPU: {fname} answers {nbad} of {n} sampled inputs differently from the record "
                             "(tests/golden/transc: MI355X, ROCm 7.2). math_mode fast cannot be verified bit for bit here; run with --math-mode exact, or re-record (tools/transc_study/study.py)")


def stand_in_outputs(cfg, blob, x0, xref, keys, u0, s0):
    """bench.py's dry run (SDEMPC_BENCH_DRY=1: the rank plumbing on a machine without a GPU): the outputs "of the GPU" that the ranks then check —
    the oracle stands in for the device; (uopt, xevol, info) arrays of the batch"""
    orc = cpu_oracle()
    O = orc.Oracle(cfg, blob)
    return [np.stack(a) for a in zip(*[O.solve(x0[i], xref[i], orc.noise_from_key(keys[i], cfg.num_particles, cfg.horizon), u0[i], s0)[:3] for i in range(len(x0))])]
