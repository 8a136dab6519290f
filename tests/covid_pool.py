"""Test infrastructure: runs oracle/covid_oracle.py (the batched NumPy restatement of the reference's COVID step)
over many replicas in blocks, on worker SUBPROCESSES (plain `python covid_pool.py job.npz`, each with a timeout) --
the oracle materialises the reference's [n, filters, 600] signal tensor per replica and step, so 8192 replicas at once
would need ~10 GB per step, and the parent holds a HIP context that must not be forked."""
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_block(cfg, acts_a, acts_p, check_at):
    from test_covid_golden import make_oracle

    E = acts_a.shape[1]
    o = make_oracle(cfg, n_envs=E)
    o.reset()
    rew_a, rew_p, state, obs = [], [], {}, {}
    for k in range(1, acts_a.shape[0] + 1):
        ob = o.step(acts_a[k - 1], acts_p[k - 1])
        rew_a.append(np.asarray(o.rew_a, np.float64).copy())
        rew_p.append(np.asarray(o.rew_p, np.float64).copy())
        if k in check_at:
            state[k] = {n: np.array(v) for n, v in o.state().items()}
            obs[k] = {n: np.array(v) for n, v in ob.items()}
    return np.stack(rew_a), np.stack(rew_p), state, obs


def _worker(job_path):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    with open(job_path, "rb") as f:
        job = pickle.load(f)
    out = [_run_block(job["cfg"], a, p, job["check_at"]) for a, p in job["blocks"]]
    with open(job_path + ".out", "wb") as f:
        pickle.dump(out, f, protocol=4)


def run_blocks(cfg, acts_a, acts_p, block, check_at, workers=8, timeout=900):
    """acts_a [T, E, n], acts_p [T, E] -> {"rew_a": [T, E, n], "rew_p": [T, E], "state": {day: {...}}, "obs": {...}}"""
    E = acts_a.shape[1]
    blocks = [(acts_a[:, lo:lo + block], acts_p[:, lo:lo + block]) for lo in range(0, E, block)]
    workers = max(1, min(workers, len(blocks)))
    per = (len(blocks) + workers - 1) // workers
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for w in range(workers):
            mine = blocks[w * per:(w + 1) * per]
            if not mine:
                continue
            path = os.path.join(d, "job%d.pkl" % w)
            with open(path, "wb") as f:
                pickle.dump({"cfg": json.loads(json.dumps(cfg)), "check_at": tuple(check_at), "blocks": mine}, f, protocol=4)
            procs.append((path, subprocess.Popen([sys.executable, os.path.abspath(__file__), path], env=env)))
        parts = []
        try:
            for path, pr in procs:
                rc = pr.wait(timeout=timeout)
                assert rc == 0, "COVID oracle worker failed (rc %d)" % rc
                with open(path + ".out", "rb") as f:
                    parts.extend(pickle.load(f))
        finally:
            for _, pr in procs:
                if pr.poll() is None:
                    pr.kill()
    out = {"rew_a": np.concatenate([p[0] for p in parts], axis=1), "rew_p": np.concatenate([p[1] for p in parts], axis=1),
           "state": {}, "obs": {}}
    for k in check_at:
        out["state"][k] = {n: np.concatenate([p[2][k][n] for p in parts], axis=0) for n in parts[0][2][k]}
        out["obs"][k] = {n: np.concatenate([p[3][k][n] for p in parts], axis=0) for n in parts[0][3][k]}
    return out


if __name__ == "__main__":
    _worker(sys.argv[1])
