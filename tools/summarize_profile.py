#!/usr/bin/env python3
"""Turn a rocprofv3 output directory (gpurun_out/<run>) into the committed summaries under profiles/.

usage: tools/summarize_profile.py gpurun_out/r01b r01 [--batch 2048 --config c2_iris_traj_h50_p128.yaml]
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, updates profiles/pmc_traffic.json.
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: separate --pmc passes; FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request -> doubled (checked on this
access pattern with the rollout kernel, whose only traffic is the known noise tensor: cal_* passes)."""
import argparse, collections, csv, glob, json, os, shutil, sys

ap = argparse.ArgumentParser()
ap.add_argument("src"); ap.add_argument("tag")
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--config", default="c2_iris_traj_h50_p128.yaml")
ap.add_argument("--mfma-per-eval", type=float, default=0.0, help="f32 MFMAs of one forward sweep of one instance (C2: 2 pairs x 50 steps x 44 = 4400); enables the SQ sum check")
ap.add_argument("--mfma-per-adjoint", type=float, default=0.0, help="matrix instructions of one ADJOINT sweep of one instance when they differ from a forward sweep's (f32x3 / fast: forward 2 x 50 x 28 = 2800, adjoint 2 x 50 x 36 = 3600)")
ap.add_argument("--mlp-dtype", default="f32x3", help="arithmetic of the profiled launches (key of profiles/pmc_traffic.json)")
ap.add_argument("--math-mode", default="fast", help="math mode of the profiled launches (key of profiles/pmc_traffic.json)")
ap.add_argument("--sq-batch", type=int, default=0, help="batch of the SQ_* / GRBM passes (pmc_3, pmc_4) when it differs from --batch (tools/profile_round.sh)")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def counters(pattern, kname):
    acc = collections.defaultdict(float); calls = collections.defaultdict(int)
    files = collections.defaultdict(list)
    for f in glob.glob(os.path.join(a.src, pattern, "*", "*_counter_collection.csv")):
        files[os.path.dirname(f)].append(f)
    for d, fs in files.items():       # gpurun MERGES a call's files into gpurun_out/: an earlier call's passes may still lie beside the new ones — newest only
        if len(fs) > 1:
            print(f"note: {len(fs)} runs in {d}, using the newest", file=sys.stderr)
        f = max(fs, key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
    return {k: v / calls[k] for k, v in acc.items()}          # per launch

ks = glob.glob(os.path.join(a.src, "trace", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(max(ks, key=os.path.getmtime), os.path.join(out, f"{a.tag}_kernel_stats.csv"))
log = os.path.join(a.src, "bench_trace.log")
bench = None
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            bench = json.loads(line)
solve = counters("pmc_[12]", "solve") if a.sq_batch else counters("pmc_*", "solve")
if a.sq_batch:
    solve.update(counters("pmc_[3-9]", "solve"))
cal_r = counters("cal_rollout_*", "rollout"); cal_g = counters("cal_grad_*", "grad")
res = {"tag": a.tag, "batch": a.batch, "config": a.config, "solve_kernel_per_launch": solve, "bench_under_rocprof": bench}
if a.sq_batch:
    res["sq_counters_batch"] = a.sq_batch
    res["note"] = (f"FETCH_SIZE / WRITE_SIZE per launch of {a.batch} instances; every SQ_* / GRBM_* counter per launch of {a.sq_batch} instances "
                   "(at 12,288 rocprofv3's SQ sums come out 9/8 of the work the kernel's own counters report; they match at 3,072 / 4,608 / 6,144)")
KiB = 1024.0
if "FETCH_SIZE" in solve and "WRITE_SIZE" in solve:
    rd, wr = 2.0 * solve["FETCH_SIZE"] * KiB, solve["WRITE_SIZE"] * KiB
    res["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr, "per_solve": (rd + wr) / a.batch}
    tf = os.path.join(out, "pmc_traffic.json")
    rec = json.load(open(tf)) if os.path.exists(tf) else {}
    sha = os.path.join(a.src, "lib_sha.txt")          # written by tools/profile_round.sh on the GPU box: sha256 of the library the counters were taken on
    build = open(sha).read().split()[0][:16] if os.path.exists(sha) else None
    res["library_build"] = build
    key = f"{a.config}:B{a.batch}:{a.mlp_dtype}:{a.math_mode}"
    rec[key] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "source": f"profiles/{a.tag}_pmc.json", "traffic_build": build}
    if "GRBM_GUI_ACTIVE" in solve and "SQ_INSTS_VALU" in solve:       # what binds the kernel, for bench.py's roofline.valu_issue (same build gate as the traffic)
        cyc = solve["GRBM_GUI_ACTIVE"] / 8.0
        rec[key]["valu_issue"] = {"insts_per_simd": solve["SQ_INSTS_VALU"] / 1024, "cycles": cyc, "frac_at_best_issue": solve["SQ_INSTS_VALU"] / 1024 * 2.8 / cyc,
                                  "best_issue_cycles_per_inst": 2.8, "mfma_busy_frac": solve.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc,
                                  "instances_per_launch": a.sq_batch or a.batch,
                                  "source": f"profiles/{a.tag}_pmc.json: SQ_INSTS_VALU / 1024 SIMDs, GRBM_GUI_ACTIVE / 8 XCDs; 2.8 cycles per wave-instruction is the best a SIMD issues with three waves (tools/valu_probe.hip)"}
    json.dump(rec, open(tf, "w"), indent=1)
if cal_r:
    res["calibration"] = {"rollout_kernel_FETCH_SIZE_KiB": cal_r.get("FETCH_SIZE"), "rollout_kernel_WRITE_SIZE_KiB": cal_r.get("WRITE_SIZE"),
                          "grad_kernel_FETCH_SIZE_KiB": cal_g.get("FETCH_SIZE"), "grad_kernel_WRITE_SIZE_KiB": cal_g.get("WRITE_SIZE"),
                          "known_rollout_read_bytes": a.batch * 4 * 50 * 6 * 32 * 4, "known_grad_write_bytes": a.batch * 4 * (51 * 13 * 32 + 50 * 1280) * 4}
if "GRBM_GUI_ACTIVE" in solve and "SQ_INSTS_VALU" in solve:
    cyc = solve["GRBM_GUI_ACTIVE"] / 8.0
    res["derived"] = {"kernel_cycles": cyc, "valu_insts_per_simd": solve["SQ_INSTS_VALU"] / 1024, "mfma_busy_frac": solve.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc,
                      "valu_issue_frac_at_2.8cyc": solve["SQ_INSTS_VALU"] / 1024 * 2.8 / cyc}
# SQ_INSTS_MFMA is known exactly from the kernel's own work counters (bench line): a factor other than 1 means the profiled launches did
# not do the work the bench line describes (round 2: launches whose instance tickets were reset late solved 1,536 instances twice)
if bench and "SQ_INSTS_MFMA" in solve and a.mfma_per_eval:
    cfgb = bench["config"]
    adj = a.mfma_per_adjoint or a.mfma_per_eval
    exp = ((a.mfma_per_eval + adj) * cfgb["N_grad_evaluated_mean"] + a.mfma_per_eval * cfgb["N_forward_rollouts_mean"]) * (a.sq_batch or a.batch)
    f = solve["SQ_INSTS_MFMA"] / exp
    res["sq_sum_check"] = {"expected_SQ_INSTS_MFMA": exp, "measured": solve["SQ_INSTS_MFMA"], "factor": f,
                           "SQ_INSTS_VALU_per_solve_corrected": solve.get("SQ_INSTS_VALU", 0) / f / (a.sq_batch or a.batch),
                           "SQ_INSTS_LDS_per_solve_corrected": solve.get("SQ_INSTS_LDS", 0) / f / (a.sq_batch or a.batch)}
json.dump(res, open(os.path.join(out, f"{a.tag}_pmc.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k in ("hbm_bytes_per_launch", "derived", "calibration", "sq_sum_check")}, indent=1))
