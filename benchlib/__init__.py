"""Pieces of bench.py (repo root): `counts` — SURVEY.md §8(d) algorithmic bytes / flops and the rooflines; `verify` — the CPU-oracle legs (checker of
the timed launches, cpu_baseline); `power` — package power and shader clock of the node's GPUs beside the timed launches; `ranks` — the self-started
`--gpus N` launcher; `legs` — one workload on one GPU, the latency loop, BASELINE config 4, the secondary configurations. bench.py keeps argument
parsing, the timed leg and the JSON line."""
