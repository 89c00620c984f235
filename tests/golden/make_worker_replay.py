"""Generates tests/golden/worker_replay.npz: the scripted mode sequence of tests/replay.py driven through an ORACLE-backed worker
(self-generated; not reference output — the reference's solver cannot run here). Run: python tests/golden/make_worker_replay.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import replay

rec = replay.run(oracle=True)
np.savez_compressed(replay.FIXTURE, **rec)
print(replay.FIXTURE, {k: v.shape for k, v in rec.items()}, os.path.getsize(replay.FIXTURE), "bytes")
