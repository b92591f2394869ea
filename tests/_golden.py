"""Helpers shared by the parity tests: fixture loading + the gradient fingerprint used by oracle/gen_golden.py."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def meta():
    with open(os.path.join(GOLD, "meta.json")) as f:
        return json.load(f)


def summarize(g, nsamp=256):
    f = np.asarray(g).reshape(-1).astype(np.float64)
    step = max(1, f.size // nsamp)
    return np.concatenate([[f.sum(), np.abs(f).sum(), np.sqrt((f * f).sum())], f[::step][:nsamp]])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def check_grads(gold: dict, grads: dict, tol, prefix_filter=None):
    """Compare a {param_name: grad} dict with the 'g:'/'gs:' entries of a fixture.  Returns the worst rel error."""
    worst = 0.0
    n = 0
    for k, v in gold.items():
        if k.startswith("g:"):
            name = k[2:]
            got = grads[name]
            e = rel_err(got, v)
        elif k.startswith("gs:"):
            name = k[3:]
            s = summarize(grads[name])
            # norms: relative; samples: relative to the l2/sqrt(n) scale
            e = max(abs(s[1] - v[1]) / (abs(v[1]) + 1e-30), abs(s[2] - v[2]) / (abs(v[2]) + 1e-30),
                    float(np.abs(s[3:] - v[3:]).max() / (np.abs(v[3:]).max() + 1e-30)))
        else:
            continue
        if prefix_filter and not name.startswith(prefix_filter):
            continue
        assert e < tol, f"grad {name}: rel err {e:.3e} >= {tol}"
        worst = max(worst, e)
        n += 1
    assert n > 0
    return worst
