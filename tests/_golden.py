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


# ---- metrics for the bf16 production path ("relative to the maximum" hides errors in small rows) ----


def row_rel_err(got, ref, axis=-1):
    """max over rows of ||got_r - ref_r||_2 / ||ref_r||_2 (rows = slices along `axis`, i.e. one token / one output feature).
    Every row is judged against ITS OWN magnitude.  Rows whose reference norm is below 1e-6 of the largest row norm (zeroed
    attention rows, padded positions) must be small in absolute terms instead: ||got_r|| <= 1e-3 of the largest row norm."""
    a = np.asarray(got, dtype=np.float64)
    b = np.asarray(ref, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.ndim == 1:
        a, b = a[None], b[None]
    nb = np.sqrt((b * b).sum(axis))
    nd = np.sqrt(((a - b) ** 2).sum(axis))
    big = float(nb.max()) + 1e-300
    live = nb > 1e-6 * big
    worst = float((nd[live] / nb[live]).max()) if live.any() else 0.0
    if (~live).any():
        na = np.sqrt((a * a).sum(axis))
        assert float(na[~live].max()) <= 1e-3 * big, "a row that is zero in the reference is not small"
    return worst


def cosine(got, ref):
    a = np.asarray(got, dtype=np.float64).reshape(-1)
    b = np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(a @ b / (np.sqrt(a @ a) * np.sqrt(b @ b) + 1e-300))


def record(name, **vals):
    """Append measured parity figures to gpurun_out/parity_metrics.jsonl when that directory exists (the GPU box): the numbers
    quoted in DESIGN.md section 5 come from there."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if not isinstance(v, (list, str, dict)) else v) for k, v in vals.items()}}) + "\n")
