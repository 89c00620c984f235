"""Float64 referee of the library's four f32 arithmetics — (f32 | f32x3) x (exact | fast): which is closer to the real-number solution?

The GPU reproduces each arithmetic's CPU oracle bit for bit (tests/test_gpu_parity.py), so the oracle's results ARE the GPU's. Here they are compared
with the float64 build of the same oracle (oracle/ -DORC_DOUBLE: the same statements on double, libm activations) on 64 C1-sized and 8 C2-sized
instances: ONE gradient per instance and the FULL cold-start solve. tests/golden/referee.npz holds every result (tests/golden/make_referee.py:
about 75 CPU-minutes, hence committed); this test recomputes EVERY gradient and a sample of the solves live — the fixture cannot drift from the
oracle — and asserts the referee's criteria on the whole table: neither `fast` nor `f32x3` may be further from float64 than plain `f32/exact` by
more than 1.5 x on the per-gradient figures. bench.py reports the same table for instances of its own timed batch (vs_float64)."""
import importlib.util
import os

import numpy as np
import pytest

import orc
from benchlib import referee as R
from cases import bits_differ

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_spec = importlib.util.spec_from_file_location("make_referee", os.path.join(GDIR, "make_referee.py"))
mk = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mk)
FX = dict(np.load(os.path.join(GDIR, "referee.npz")))
NTHR = min(os.cpu_count() or 1, 8)


def _stored(name):
    return {a: (FX[f"{name}_{a.replace('/', '_')}_grad"], FX[f"{name}_{a.replace('/', '_')}_gcost"], FX[f"{name}_{a.replace('/', '_')}_uopt"]) for a in mk.ARITHS}


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_every_gradient_of_the_fixture_is_what_the_oracle_computes_today(name):
    """all 64 / 8 instances, all five arithmetics: one gradient each, recomputed (f32 arithmetics bit for bit; float64 to 1e-12: libm)"""
    live = mk.compute(name, NTHR, solves=False, log=lambda *_: None)
    st = _stored(name)
    assert len(st["f64"][0]) == mk.SETS[name][1] and mk.SETS["c1"][1] >= 64 and mk.SETS["c2"][1] >= 8
    for a in mk.ARITHS:
        g, c, _ = live[a]
        if a == "f64":
            np.testing.assert_allclose(g, st[a][0], rtol=1e-12, atol=1e-12 * np.abs(st[a][0]).max())
            np.testing.assert_allclose(c, st[a][1], rtol=1e-12)
        else:
            assert bits_differ(g.astype(np.float32), st[a][0].astype(np.float32)) == 0 and np.array_equal(c, st[a][1]), a


@pytest.mark.parametrize("name,which", [("c1", range(8)), ("c2", [0])])
def test_a_sample_of_the_full_solves_is_what_the_oracle_computes_today(name, which):
    """eight C1-sized instances in all five arithmetics; one C2-sized instance in float64, f32/exact and f32/fast (the f32x3 solves of a C2-sized
    instance take the matrix-instruction model two minutes each: they are pinned by the GPU suite and by bench.py's own checks of its timed launch)"""
    which = list(which)
    cfg, x0, xref, keys, u0, ug = mk.problems(name)
    model = mk.synthetic_iris()
    ariths = mk.ARITHS if name == "c1" else ["f64", "f32/exact", "f32/fast"]
    st = _stored(name)
    jobs = [(a, i) for a in ariths for i in which]
    if name == "c2":
        orc.set_threads(max(1, NTHR // len(jobs)))          # three jobs: the cores they leave idle go into each solve's particle loops (same bits)
    try:
        res = R.run_threads([(lambda a=a, i=i: mk.oracle_for(cfg, model, a).solve(x0[i], xref[i], orc.noise_from_key(keys[i], cfg.num_particles, cfg.horizon),
                                                                                 u0[i], cfg.ls_init_stepsize)[0]) for a, i in jobs], NTHR)
    finally:
        orc.set_threads(1)
    for (a, i), u in zip(jobs, res):
        if a == "f64":           # (a float64 solve returns float32 controls; another libm could flip a line-search decision: regenerate the fixture then)
            np.testing.assert_allclose(u, st[a][2][i], rtol=0, atol=1e-6, err_msg=f"{name}[{i}] float64 solve differs from the fixture: python tests/golden/make_referee.py")
        else:
            assert bits_differ(u, st[a][2][i]) == 0, (name, a, i)


def test_no_arithmetic_is_further_from_float64_than_the_plain_f32_evaluation():
    """The referee's verdict, on the whole fixture. Per gradient: rms error over every entry and the worst entry, relative to the gradient's largest
    entry — `fast` and `f32x3` within 1.5 x of f32/exact. Per full solve: reported (the table is profiles/r5_referee.json and SPEC.md §10d);
    asserted only that every arithmetic keeps at least as many instances within the north star's 1e-4 of the float64 solve as f32/exact does, minus
    one instance in eight (a flipped line-search decision is a coin toss that any rounding loses sometimes, the plain f32 evaluation included)."""
    for name in ("c1", "c2"):
        st = _stored(name)
        n = len(st["f64"][0])
        grad_rows = {a: [R.gradient_error(st[a][0][i], st["f64"][0][i], st[a][1][i], st["f64"][1][i]) for i in range(n)] for a in mk.ARITHS[1:]}
        solve_rows = {a: [R.solve_error(st[a][2][i], st["f64"][2][i]) for i in range(n)] for a in mk.ARITHS[1:]}
        table = R.summarize(grad_rows, solve_rows)
        ratios = R.ratios_to(table)
        print(name, {k: {q: round(v, 3) for q, v in r.items()} for k, r in ratios.items()},
              {k: (row["solves_within_1e-4_of_float64"], row["max_abs_du_median"], row["max_abs_du_worst"]) for k, row in table.items()})
        base = table["f32/exact"]
        assert 1e-9 < base["grad_rms_rel"] < 1e-5 and base["grad_max_rel_worst"] < 1e-4          # f32 gradients are f32-accurate, and not float64 by accident
        for a, r in ratios.items():
            assert r["grad_rms_rel"] <= 1.5 and r["grad_max_rel_worst"] <= 1.5, (name, a, r)
            assert table[a]["solves_within_1e-4_of_float64"] >= base["solves_within_1e-4_of_float64"] - 0.125 - 1e-9, (name, a)
