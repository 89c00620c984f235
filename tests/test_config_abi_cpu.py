"""CPU tests: YAML schema parsing, model blob layout and the C-ABI library surface (no compute calls)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from cases import CDIR, ROOT
from sde4mbrl_px4_amd import MPCConfig, load_mpc_config, synthetic_hexa, synthetic_iris
from sde4mbrl_px4_amd import _abi

REF_LAUNCH = "/root/reference/launch"


def test_benchmark_configs_parse():
    c2 = load_mpc_config(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"))
    assert (c2.horizon, c2.num_particles, c2.num_motors, c2.max_iter, c2.ls_maxls) == (50, 128, 4, 200, 4)
    assert c2.perr == [100.0, 100.0, 200.0] and c2.qerr == [1.0, 1.0, 100.0] and c2.u_slew_constr is None
    assert c2.moment_scale is None and c2.ls_reset_option == "increase" and c2.ls_max_stepsize == 1.0
    np.testing.assert_allclose(c2.time_steps, np.full(50, 0.05, np.float32))
    c1 = load_mpc_config(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"))
    assert c1.max_iter == 100 and c1.ls_max_stepsize == 10.0 and c1.u_slew_constr[1] == [-26.0, 0.32] and c1.uerr == pytest.approx(0.01)
    c3 = load_mpc_config(os.path.join(CDIR, "c3_hexa_traj_h50_p256.yaml"))
    assert c3.num_motors == 6 and c3.uref == pytest.approx([0.42] * 6) and c3.num_particles == 256
    c5 = load_mpc_config(os.path.join(CDIR, "c5_iris_traj_h200_p1024.yaml"))
    assert (c5.horizon, c5.num_particles) == (200, 1024)


@pytest.mark.skipif(not os.path.isdir(REF_LAUNCH), reason="reference tree not present (GPU box)")
def test_reference_yaml_files_parse_verbatim():
    """Every shipped MPC YAML of the reference parses; shipped sizes are H=20, P=1."""
    files = sorted(glob.glob(os.path.join(REF_LAUNCH, "*_mpc.yaml")))
    assert len(files) == 6
    for f in files:
        c = load_mpc_config(f)
        assert c.horizon == 20 and c.num_particles == 1 and c.num_motors in (4, 6)
        assert c.trajectory_path is None or c.trajectory_path.endswith(".csv")
        cfg, keep = c.to_cfg()
        assert cfg.struct_size == C.sizeof(_abi.SdempcCfg) and cfg.horizon == 20
    iris = load_mpc_config(os.path.join(REF_LAUNCH, "iris_sitl_traj_mpc.yaml"))
    ours = load_mpc_config(os.path.join(CDIR, "iris_traj_shipped_h20_p1.yaml"))
    for k in ("uref", "uerr", "perr", "verr", "qerr", "werr", "res_mult", "u_slew_coeff", "horizon", "num_particles", "max_iter",
              "beta_init", "atol", "rtol", "ls_init_stepsize", "ls_max_stepsize", "ls_coef", "ls_decrease_factor", "ls_increase_factor",
              "ls_maxls", "input_bound", "discount", "short_step_dt"):
        assert getattr(iris, k) == getattr(ours, k), k


def test_time_grid_and_validation():
    c = MPCConfig(horizon=7, num_short_dt=3, short_step_dt=0.02, long_step_dt=0.1)
    np.testing.assert_allclose(c.time_steps, [0.02] * 3 + [0.1] * 4)
    with pytest.raises(ValueError):
        MPCConfig(ls_reset_option="bogus").to_cfg()
    with pytest.raises(ValueError):
        MPCConfig(uref=[0.7] * 3).to_cfg()
    cfg, _ = MPCConfig(enforce_ubound=False).to_cfg()
    assert cfg.u_lo[0] < -1e30 and cfg.u_hi[0] > 1e30


def test_model_blob_layout():
    for model in (synthetic_iris(), synthetic_hexa()):
        blob = model.to_blob()
        assert len(blob) == 4 * (_abi.BLOB_HEADER_INTS + _abi.BLOB_FLOATS)
        hd = np.frombuffer(blob[:64], np.int32)
        assert hd[0] == _abi.BLOB_MAGIC and hd[2] == model.num_motors and hd[3] == 32
        f = np.frombuffer(blob[64:], np.float32)
        assert f[0] == np.float32(1.0) / np.float32(model.mass)
        np.testing.assert_array_equal(f[56:56 + 384].reshape(64, 6), model.W1z)
        np.testing.assert_array_equal(f[56 + 384 + 64:56 + 384 + 64 + 256].reshape(32, 8)[:, :model.num_motors], model.W1u)
    # hover equilibrium of the synthetic vehicles: thrust at uref equals weight (2 % tolerance)
    m = synthetic_iris()
    T = 4 * (m.thrust_poly[0] * 0.71 ** 2 + m.thrust_poly[1] * 0.71)
    assert abs(T - m.mass * m.grav) < 0.02 * m.mass * m.grav


def test_model_npz_round_trip(tmp_path):
    from sde4mbrl_px4_amd import RotorSDEModel
    m = synthetic_hexa(3)
    f = str(tmp_path / "model.npz")
    m.save_npz(f)
    assert RotorSDEModel.load_npz(f).to_blob() == m.to_blob()
    d = dict(np.load(f))
    d["W2"] = np.zeros((32, 31), np.float32)
    np.savez(f, **d)
    with pytest.raises(ValueError, match="W2"):
        RotorSDEModel.load_npz(f)


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sdempc.h")).read()
    return sorted(set(re.findall(r"\b(sdempc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    syms = _header_symbols()
    assert set(syms) == set(_abi.EXPORTED_SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), s
    hdr = open(os.path.join(ROOT, "include", "sdempc.h")).read()
    assert lib.sdempc_abi_version() == _abi.ABI_VERSION == int(re.search(r"#define SDEMPC_ABI_VERSION (\d+)", hdr).group(1))


def test_handle_lifecycle_and_errors_without_gpu():
    lib = _abi.load_library()
    cfg_py = load_mpc_config(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"))
    cfg, keep = cfg_py.to_cfg()
    blob = synthetic_iris().to_blob()
    buf = C.create_string_buffer(blob, len(blob))
    h = C.c_void_p()
    assert lib.sdempc_create(C.byref(cfg), buf, len(blob), 4, C.byref(h)) == 0
    # m_reset is host-only: hover guess + initial telemetry (sde_control.py:706-707 reads .yk)
    yk = np.zeros((20, 4), np.float32)
    info = _abi.SdempcInfo()
    x = np.zeros(13, np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.sdempc_reset(h, x.ctypes.data_as(fp), x.ctypes.data_as(fp), yk.ctypes.data_as(fp), C.byref(info)) == 0
    np.testing.assert_allclose(yk, 0.71)
    assert info.stepsize == pytest.approx(0.01) and info.num_steps == 0
    assert lib.sdempc_noise_dev_floats(h, 3) == 3 * 1 * 20 * 6 * 32
    assert lib.sdempc_traj_dev_floats(h, 2) == 2 * 1 * 21 * 13 * 32
    # capacity is checked before any device work
    dummy = np.zeros(8, np.float32).ctypes.data_as(fp)
    assert lib.sdempc_rollout_batch(h, 5, dummy, dummy, dummy, dummy, dummy, None, None) == -5
    assert b"max_batch" in lib.sdempc_last_error(h)
    assert lib.sdempc_rollout_batch(h, 0, dummy, dummy, dummy, dummy, dummy, None, None) == -1          # empty batch
    lib.sdempc_destroy(h)
    # horizon limits: one workgroup's LDS (160 KiB) bounds H; (3550 + 81 H) floats for 4 motors: 450 still fits, 480 does not
    for H, ok in ((256, True), (450, True), (480, False), (0, False)):
        cl, k2 = cfg_py.replace(horizon=H, num_short_dt=H).to_cfg() if H > 0 else (None, None)
        if H == 0:
            cl = _abi.SdempcCfg.from_buffer_copy(cfg)
            cl.horizon = 0
        hh = C.c_void_p()
        rc = lib.sdempc_create(C.byref(cl), buf, len(blob), 1, C.byref(hh))
        assert (rc == 0) == ok, (H, rc, lib.sdempc_last_error(None))
        if rc == 0:
            lib.sdempc_destroy(hh)
    # bad arguments
    h2 = C.c_void_p()
    bad = _abi.SdempcCfg.from_buffer_copy(cfg)
    bad.struct_size = 12
    assert lib.sdempc_create(C.byref(bad), buf, len(blob), 1, C.byref(h2)) == -1
    assert lib.sdempc_create(C.byref(cfg), buf, 100, 1, C.byref(h2)) == -2
    hexa = synthetic_hexa().to_blob()
    hb = C.create_string_buffer(hexa, len(hexa))
    assert lib.sdempc_create(C.byref(cfg), hb, len(hexa), 1, C.byref(h2)) == -1     # 4-motor cfg, 6-motor model
    assert b"num_motors" in lib.sdempc_last_error(None)
    corrupted = bytearray(blob)
    corrupted[0] ^= 0xFF
    cb = C.create_string_buffer(bytes(corrupted), len(corrupted))
    assert lib.sdempc_create(C.byref(cfg), cb, len(blob), 1, C.byref(h2)) == -2


def test_no_exception_crosses_the_c_abi():
    """include/sdempc.h promises "never aborts or throws" (an exception would end the reference's mpc_process silently,
    sde_control.py:365-419): arguments that make the host tables unallocatable come back as a code with a message."""
    lib = _abi.load_library()
    cfg_py = load_mpc_config(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"))
    blob = synthetic_iris().to_blob()
    buf = C.create_string_buffer(blob, len(blob))
    h = C.c_void_p()
    # the largest horizon the LDS check lets through, the largest batch an int32 can name: creation itself is host-only and must succeed
    # or fail with a code; the first device call then has to refuse the workspace (no GPU here: EDEVICE either way), never crash
    cl, keep = cfg_py.replace(horizon=450, num_short_dt=450).to_cfg()
    rc = lib.sdempc_create(C.byref(cl), buf, len(blob), 2**31 - 1, C.byref(h))
    assert rc in (0, -4), (rc, lib.sdempc_last_error(None))
    if rc == 0:
        fp = C.POINTER(C.c_float)
        dummy = np.zeros(8, np.float32).ctypes.data_as(fp)
        assert lib.sdempc_rollout_batch(h, 1, dummy, dummy, dummy, dummy, dummy, None, None) in (-3, -4)
        assert lib.sdempc_last_error(h)
        lib.sdempc_destroy(h)
    # a momentum table of 2^31 + 1 entries: std::length_error / bad_alloc inside the library -> SDEMPC_EINVAL / SDEMPC_ENOMEM, not a crash
    for max_iter in (2**31 - 1, 2**31 - 2, 10**7 + 1):
        bad = _abi.SdempcCfg.from_buffer_copy(cl)
        bad.max_iter = max_iter
        h2 = C.c_void_p()
        rc = lib.sdempc_create(C.byref(bad), buf, len(blob), 1, C.byref(h2))
        assert rc in (-1, -4) and not h2.value, (max_iter, rc)
        assert lib.sdempc_last_error(None)
    # 10^7 iterations is the documented limit and must still be accepted (40 MB of momentum table)
    ok = _abi.SdempcCfg.from_buffer_copy(cl)
    ok.max_iter = 10**7
    h3 = C.c_void_p()
    assert lib.sdempc_create(C.byref(ok), buf, len(blob), 1, C.byref(h3)) == 0
    lib.sdempc_destroy(h3)
    # the guard is in every entry point: the source has no extern "C" function returning a code whose body is not inside guarded()
    src = open(os.path.join(ROOT, "sde4mbrl_px4_amd", "csrc", "sdempc_api.cpp")).read()
    bodies = re.findall(r"\n(?:int|int32_t) (sdempc_\w+)\([^{]*\{\n(.*?)\n", src, re.S)
    unguarded = [n for n, first in bodies if "guarded(" not in first and n not in ("sdempc_abi_version", "sdempc_device_ready", "sdempc_layout_fallbacks")]
    assert not unguarded, unguarded


def test_no_cpu_fallback_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sde4mbrl_px4_amd.solver import SdempcError, SdeMpcSolver
    from sde4mbrl_px4_amd import workload as W
    cfg = MPCConfig(horizon=4, num_short_dt=4, num_particles=2, max_iter=1)
    S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=1)
    with pytest.raises(SdempcError, match="no HIP device|failed"):
        S.rollout(W.random_initial_states(1), np.full((1, 4, 4), 0.7, np.float32), W.reference_window(0, cfg.time_steps)[None], W.make_noise(1, 2, 4))
    S.close()


def test_entry_scripts_are_valid_python():
    import ast
    for name in ("bench.py", "benchlib/counts.py", "benchlib/verify.py", "benchlib/power.py", "benchlib/ranks.py", "benchlib/legs.py", "__graft_entry__.py",
                 "tools/prof_solve.py", "tools/summarize_profile.py", "tests/golden/make_golden.py"):
        ast.parse(open(os.path.join(ROOT, name)).read(), filename=name)
    import bench      # noqa: F401  (imports benchlib; no GPU is touched at import time)
    from benchlib import counts, power, verify
    cfg = load_mpc_config(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"))
    nbytes, nflops, b_grad, b_ls = counts.algorithmic_counts(cfg, 200, 375)
    assert b_grad == 990364 and b_ls == 157052                      # SURVEY.md §8d formulas with n_w = 6
    assert nflops == 3520 * 128 * 50 * (2 * 200 + 375 + 2)
    assert counts.checkpoint_bytes(cfg, 200) == 200 * 2 * 4 * 50 * 1280 * 4
    assert verify.effective_cores() >= 1
    # the background checker of timed launches: a live queue (jobs added after the start are picked up), mismatching words are counted
    import orc
    from sde4mbrl_px4_amd import MPCConfig as _Cfg, synthetic_iris as _iris, prng as _prng
    from sde4mbrl_px4_amd import workload as _W
    vc = _Cfg(horizon=6, num_short_dt=6, num_particles=16, u_slew_coeff=1.0, max_iter=3, max_no_improvement_iter=3)
    vb = _iris().to_blob()
    vx0 = _W.random_initial_states(5, 0); vxr = np.stack([_W.reference_window(0.0, vc.time_steps)] * 5)
    vk = _prng.split(_prng.PRNGKey(1), 5); vu0 = np.tile(np.asarray(vc.uref, np.float32), (5, 6, 1))
    vO = orc.Oracle(vc, vb)
    vo = [vO.solve(vx0[b], vxr[b], orc.noise_from_key(vk[b], 16, 6), vu0[b], 0.01)[:3] for b in range(5)]
    got = tuple(np.stack([o[k] for o in vo]) for k in range(3))
    V = verify.Verifier(3)
    V.add("main", vc, vb, [0, 2], vx0, vxr, vk, vu0, 0.01, got)
    V.start()
    wrong = (got[0].copy(), got[1], got[2]); wrong[0][4, 0, 0] += 1e-3
    V.add("late", vc, vb, [1, 4], vx0, vxr, vk, vu0, 0.01, wrong)
    V.join()
    assert V.results["main"] == dict(V.results["main"], bad_words=0, done=2) and V.results["late"]["bad_words"] == 1 and V.results["late"]["done"] == 2
    assert verify.Verifier(2).join() == 0.0                                   # never started (--verify 0)
    assert V.incomplete() == {} and V.errors == []
    # a worker that dies leaves its instance unchecked, which bench.py turns into a non-zero exit (the exception is kept for the message)
    Vd = verify.Verifier(1)
    Vd.add("main", vc, vb, [0], vx0, vxr, vk, vu0, 0.01, (got[0], got[1][:, :2], got[2]))          # a malformed xevol: the comparison raises
    Vd.start(); Vd.join()
    assert Vd.incomplete() == {"main": (0, 1)} and len(Vd.errors) == 1
    # the power / clock sampler beside the timed launches (rank 0, every GPU of the job in one pass). No amdgpu card in sysfs here, so this is its
    # rocm-smi path: ONE child per second for all devices, only samples inside the timed region count, a missing tool leaves nulls
    import subprocess, time, types
    assert power.amdgpu_cards() == [] or all(len(c) == 3 for c in power.amdgpu_cards())
    calls = []
    def fake(cmd, **kw):
        calls.append(cmd)
        if "--showmaxpower" in cmd:
            return types.SimpleNamespace(stdout="GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n", returncode=0)
        return types.SimpleNamespace(stdout="".join(f"GPU[{g}]\t\t: sclk clock level: 1: ({1896 - 10 * g}Mhz)\nGPU[{g}]\t\t: Current Socket Graphics Package Power (W): {1399.0 - g}\n" for g in range(4)), returncode=0)
    real, real_cards = subprocess.run, power.amdgpu_cards
    try:
        subprocess.run = fake
        power.amdgpu_cards = lambda: []
        ps = power.PowerSampler([0, 1, 2, 3])
        t0 = time.perf_counter() - 1.5
        ps.start(); time.sleep(2.6)
        r = ps.summary(t0, time.perf_counter())
        assert r["power_cap_w"] == 1400.0 and r["package_power_w_median"] == 1399.0 and r["sclk_mhz_median"] == 1896.0 and r["samples"] >= 1
        assert r["over_gpus"]["gpus_sampled"] == 4 and r["over_gpus"]["sclk_mhz_median_min"] == 1866.0 and r["over_gpus"]["package_power_w_median_min"] == 1396.0
        assert all("-d" not in c for c in calls if "--showpower" in c)                  # one call covers all devices
        def missing(cmd, **kw): raise FileNotFoundError("rocm-smi")
        subprocess.run = missing
        ps = power.PowerSampler([0]); ps.start(); time.sleep(0.2)
        assert ps.summary(0.0, 1e9)["samples"] == 0
    finally:
        subprocess.run, power.amdgpu_cards = real, real_cards


def test_product_and_tools_never_touch_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"liborc|import orc\b|from orc\b|oracle/|orc_[a-z]+\(")
    offenders = []
    for f in glob.glob(os.path.join(root, "sde4mbrl_px4_amd", "**", "*"), recursive=True) + glob.glob(os.path.join(root, "tools", "*")):
        if os.path.isfile(f) and f.endswith((".py", ".cpp", ".hip", ".h", ".sh", "Makefile")):
            for i, line in enumerate(open(f, errors="ignore")):
                if pat.search(line) and not line.lstrip().startswith(("#", "//", "*", '"')) and "oracle/prng_oracle.c" not in line:
                    offenders.append(f"{os.path.relpath(f, root)}:{i + 1}: {line.strip()[:100]}")
    assert not offenders, "\n".join(offenders)
    # bench.py's CPU legs live in benchlib/verify.py: the oracle is loaded in exactly one function there, called only by the checker of the timed
    # outputs and by cpu_baseline; nothing else of bench.py / benchlib mentions it, and nothing of it sits inside the timed region
    src = open(os.path.join(root, "benchlib", "verify.py")).read()
    assert src.count("import orc") == 1 and src.split("import orc")[0].rsplit("\ndef ", 1)[-1].startswith("cpu_oracle()")
    callers = {blk.split("(", 1)[0].strip() for blk in re.split(r"\n(?:    )?def ", src)[1:] if "cpu_oracle()" in blk.split("\n", 1)[1]}
    # (start: Verifier.start — checks, float64 referee, cpu_baseline; calibrate: one three-iteration solve that sizes the sample; instruction_model_check: the
    # recorded instruction answers against the GPU's at start-up; stand_in_outputs: the GPU-less dry run of the rank plumbing)
    assert callers == {"cpu_solve_instances", "cpu_c1_single_solve_ms", "start", "calibrate_checker_seconds", "instruction_model_check", "stand_in_outputs"}, callers
    for other in ("bench.py", "benchlib/counts.py", "benchlib/power.py", "benchlib/ranks.py", "benchlib/legs.py", "benchlib/referee.py"):
        assert not re.search(r"import orc\b|liborc|cpu_oracle\(", open(os.path.join(root, other)).read()), other
    main_src = open(os.path.join(root, "bench.py")).read()
    timed = main_src.split("t0 = time.perf_counter()")[1].split("t1 = time.perf_counter()")[0]
    assert "cpu_" not in timed and "orc" not in timed and "Verifier" not in timed


def test_state_constr_section_parses_in_penalty_form(tmp_path):
    """The state_constr section every reference YAML ships commented out (iris_sitl_traj_mpc.yaml:16-29): uncommented it parses into the
    penalty form of SPEC.md §5.3; the slack-variable form (slack_proximal: True) is refused with a clear message."""
    import yaml
    base = yaml.safe_load(open(os.path.join(CDIR, "iris_traj_shipped_h20_p1.yaml")))
    base["state_constr"] = {"state_id": [3, 4, 5, 10, 11, 12], "state_penalty": [10.0, 10.0, 20.0, 10.0, 10.0, 10.0],
                            "slack_scaling": [3.0] * 6, "state_bound": [[-0.5, 0.5], [-0.5, 0.5], [-0.4, 0.7], [-0.8, 0.8], [-0.8, 0.8], [-0.7, 0.7]],
                            "slack_proximal": False, "constr_pen": 0.1}
    f = tmp_path / "sc.yaml"
    f.write_text(yaml.safe_dump(base))
    c = load_mpc_config(str(f))
    assert c.state_id == [3, 4, 5, 10, 11, 12] and c.constr_pen == pytest.approx(0.1) and c.state_bound[2] == [-0.4, 0.7]
    cfg, _ = c.to_cfg()
    assert cfg.num_state_constr == 6 and list(cfg.state_id)[:6] == [3, 4, 5, 10, 11, 12]
    assert cfg.state_w[2] == np.float32(np.float32(20.0) * np.float32(0.1)) and cfg.state_lo[2] == np.float32(-0.4) and cfg.state_hi[5] == np.float32(0.7)
    base["state_constr"]["slack_proximal"] = True
    f.write_text(yaml.safe_dump(base))
    with pytest.raises(NotImplementedError, match="slack_proximal"):
        load_mpc_config(str(f))
    with pytest.raises(ValueError, match="ascending"):
        c.replace(state_id=[4, 3, 5, 10, 11, 12]).to_cfg()
    # the C ABI validates the same
    from sde4mbrl_px4_amd import synthetic_iris
    from sde4mbrl_px4_amd.solver import SdeMpcSolver
    S = SdeMpcSolver(c, synthetic_iris(), max_batch=1)
    S.close()
