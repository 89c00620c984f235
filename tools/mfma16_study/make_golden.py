"""tests/golden/mfma16_{f16,bf16}.npz: a committed sample of what the HARDWARE answered (tools/mfma16_study/mfma16_probe on MI355X) for the
tiles of gen_cases.py / gen_cases2.py — 2,000 experiments per family and operand type, every tile of every family represented.
These are fixtures of the hardware's behaviour, not of this repository's code: tests/test_mfma16_model_cpu.py replays them through
oracle/mfma16_model.c.   python tools/mfma16_study/make_golden.py CASEDIR:OUTDIR [CASEDIR:OUTDIR ...]"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mfmalib import load_cases, load_out

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
per_family = 2000
acc = {"f16": {}, "bf16": {}}
for pair in sys.argv[1:]:
    cdir, odir = pair.split(":")
    for f in sorted(glob.glob(os.path.join(cdir, "*.bin"))):
        name = os.path.basename(f)[:-4]
        dtn, fam = name.split("_", 1)
        out = os.path.join(odir, name + ".out")
        if not os.path.exists(out):
            continue
        dt, A, B, C = load_cases(f)
        T = A.shape[0]
        D = load_out(out, T)
        import zlib
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        t = rng.integers(0, T, per_family); i = rng.integers(0, 32, per_family); j = rng.integers(0, 32, per_family)
        t[:min(T, per_family)] = np.arange(min(T, per_family))          # every tile at least once
        acc[dtn][fam] = dict(a=A[t, i, :].astype(np.uint16), b=B[t, :, j].astype(np.uint16), c=C[t, i, j].astype(np.uint32), d=D[t, i, j].astype(np.uint32))
for dtn, fams in acc.items():
    if not fams:
        continue
    names = sorted(fams)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"mfma16_{dtn}.npz"),
                        family=np.concatenate([np.full(len(fams[n]["c"]), k, np.int16) for k, n in enumerate(names)]), family_names=np.array(names),
                        a=np.concatenate([fams[n]["a"] for n in names]), b=np.concatenate([fams[n]["b"] for n in names]),
                        c=np.concatenate([fams[n]["c"] for n in names]), d=np.concatenate([fams[n]["d"] for n in names]))
    print(dtn, {n: len(fams[n]["c"]) for n in names})
