"""`bench.py --gpus N` without a launcher: the parent — before anything touches the GPU — starts the N ranks as child processes (rendezvous on
127.0.0.1, a free port), relays rank 0's JSON line, watches every child, and when one fails says WHICH rank failed and with what."""
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time


def spawn_ranks(n, script, argv, watchdog_s=None):
    """Start ranks 0..n-1 of `script argv`; return the exit code for the parent (0 only if every rank returned 0).
    Every rank's stderr (and, for ranks >= 1, stdout) goes to the parent's stderr as it comes AND to a per-rank log whose tail is quoted in the
    summary line when that rank fails — an RCCL initialisation failure on rank 5 is then named in the parent's own last line, not only somewhere in
    an interleaved stream. watchdog_s: kill everything after that many seconds (tests)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, logs, pumps = [], [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        log = tempfile.TemporaryFile(mode="w+")
        procs.append(p); logs.append(log)

        def pump(p=p, log=log, r=r):
            for line in iter(p.stderr.readline, b""):
                s = line.decode(errors="replace")
                log.write(s)
                sys.stderr.write(s if n == 1 else f"[rank {r}] {s}")
            sys.stderr.flush()
        t = threading.Thread(target=pump, daemon=True); t.start(); pumps.append(t)
    # rank 0's stdout is drained by a thread so that the parent can watch every child: a rank that dies (no GPU for it, bad install)
    # would otherwise leave the others waiting in the rendezvous / a barrier until the collective timeout
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed, first_bad, t0 = 0, None, time.time()
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        timed_out = watchdog_s is not None and time.time() - t0 > watchdog_s
        if bad or timed_out:
            first_bad = bad[0] if bad else (None, None)
            failed = (abs(bad[0][1]) or 1) if bad else 124
            for p in procs:             # exactly the children started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    rd.join(timeout=10)
    rcs = [p.wait() for p in procs]
    [t.join(timeout=5) for t in pumps]
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    rc = failed or max(abs(c) for c in rcs)
    if rc:
        if first_bad is None:
            first_bad = next(((r, c) for r, c in enumerate(rcs) if c), (None, None))
        r = first_bad[0]
        tail = ""
        if r is not None:
            logs[r].seek(0)
            tail = " | ".join(l.strip() for l in logs[r].read().splitlines()[-6:] if l.strip())
        what = f"rank {r} exited with code {first_bad[1]}" if r is not None else f"no rank finished within {watchdog_s} s"
        print(f"bench.py: {what}; the other ranks were stopped (exit codes by rank: {rcs}). Last lines of that rank's stderr: {tail[-1500:]}", file=sys.stderr, flush=True)
    for log in logs:
        log.close()
    return rc


def fingerprint(data) -> int:
    """64 bits of sha256 as a signed integer (what fits an int64 collective)"""
    import hashlib
    if isinstance(data, str):
        data = data.encode()
    return int.from_bytes(hashlib.sha256(bytes(data)).digest()[:8], "little", signed=True)


def ranks_that_disagree(rows):
    """rows: per-rank lists of fingerprints (gather_int64). Returns {column: [ranks whose value is not the one most ranks hold]} for the columns
    that are not unanimous (ties: the value rank 0 holds counts as the common one)."""
    bad = {}
    for c in range(len(rows[0])):
        col = [r[c] for r in rows]
        if len(set(col)) > 1:
            counts = {}
            for v in col:
                counts[v] = counts.get(v, 0) + 1
            best = max(counts.values())
            common = col[0] if counts[col[0]] == best else next(v for v in col if counts[v] == best)
            bad[c] = [r for r, v in enumerate(col) if v != common]
    return bad


def require_same_on_every_rank(named, device=None, force=False):
    """named: {what: bytes | str} held by THIS rank (the model blob it received, the build of the library it loaded). Every rank compares every
    rank's fingerprints; a rank holding something else than the others is an error of the whole run, raised on every rank with the rank named."""
    from sde4mbrl_px4_amd.dist import gather_int64
    names = list(named)
    rows = gather_int64([fingerprint(named[k]) for k in names], device=device, force=force)
    bad = ranks_that_disagree(rows)
    if bad:
        what = "; ".join(f"{names[c]}: rank(s) {rs} hold another one than the other ranks" for c, rs in bad.items())
        raise SystemExit(f"bench.py: the ranks do not run the same job — {what}")
    return {k: rows[0][i] & 0xFFFFFFFFFFFFFFFF for i, k in enumerate(names)}
