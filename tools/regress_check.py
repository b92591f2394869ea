"""tools/regress.sh's verdict: compares the figures of THIS regress call with the last committed one under profiles/ and exits non-zero when a
tracked figure is more than 3 % worse -- provided the two boxes are comparable (the bench line's own calibration: MFMA rate on random operands
and the 1 GiB copy rate both within 2 %); boxes of the pool differ by up to 5 % on the same code, so across unlike boxes the check only reports.
Tracked: ms per step, the in-situ FFN-shape GEMM average, the stand-alone gated block, and the per-shape cold-operand GEMM times (own kernels).
Usage: regress_check.py <tag>      (reads gpurun_out/<tag>_regress_*; baseline = newest profiles/*_regress_bench.json with another tag)"""
import glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
new_dir = os.path.join(ROOT, "gpurun_out")


def load_bench(path):
    with open(path) as f:
        txt = f.read().strip().splitlines()
    return json.loads(txt[-1])


def cold_table(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(\S.*?\S)\s+\[.*\]\s+us\s+min\s+([0-9.]+)", line)
        if m and re.search(r"\bown\b", m.group(1)):
            out[re.sub(r"\s+", " ", m.group(1))] = float(m.group(2))
    return out


new_b = load_bench(os.path.join(new_dir, tag + "_regress_bench.json"))
cands = sorted((p for p in glob.glob(os.path.join(ROOT, "profiles", "*_regress_bench.json")) if not os.path.basename(p).startswith(tag + "_")),
               key=lambda p: os.path.getmtime(p))
if not cands:
    print("regress_check: no committed baseline under profiles/*_regress_bench.json -- nothing to compare")
    sys.exit(0)
base_path = cands[-1]
base_tag = os.path.basename(base_path)[: -len("_regress_bench.json")]
old_b = load_bench(base_path)


def cal(b):
    c = b.get("calibration") or {}
    return (c.get("random_operands") or {}).get("tflops"), c.get("hbm_copy_tbps")


(n_tf, n_bw), (o_tf, o_bw) = cal(new_b), cal(old_b)
comparable = all(x for x in (n_tf, n_bw, o_tf, o_bw)) and abs(n_tf / o_tf - 1) <= 0.02 and abs(n_bw / o_bw - 1) <= 0.02
print("regress_check: this call (%s) MFMA %s TF / copy %s TB/s   vs   baseline %s: %s TF / %s TB/s  ->  %s" %
      (tag, n_tf, n_bw, base_tag, o_tf, o_bw, "comparable boxes: a regression FAILS" if comparable else "unlike boxes: report only"))
rows = [("ms per step", new_b["ms_per_step"], old_b["ms_per_step"])]
nr, orf = new_b.get("roofline") or {}, old_b.get("roofline") or {}
if nr.get("avg_us") and orf.get("avg_us"):
    rows.append(("FFN-shape GEMM in situ, us", nr["avg_us"], orf["avg_us"]))
if (nr.get("gated_block") or {}).get("ms") and (orf.get("gated_block") or {}).get("ms"):
    rows.append(("gated block fwd+bwd, ms", nr["gated_block"]["ms"], orf["gated_block"]["ms"]))
nc = cold_table(os.path.join(new_dir, tag + "_regress_gemm_cold_ab.txt"))
oc = cold_table(os.path.join(ROOT, "profiles", base_tag + "_regress_gemm_cold_ab.txt"))
for k in sorted(set(nc) & set(oc)):
    rows.append(("GEMM " + k + ", us", nc[k], oc[k]))
bad = 0
for name, new, old in rows:
    d = new / old - 1
    flag = "REGRESSION" if d > 0.03 else ("better" if d < -0.03 else "")
    bad += d > 0.03
    print("  %-44s %10.2f  (baseline %10.2f)  %+6.1f %%  %s" % (name, new, old, 100 * d, flag))
if bad and comparable:
    print("regress_check: %d tracked figure(s) more than 3 %% worse than %s on a comparable box" % (bad, base_tag))
    sys.exit(1)
print("regress_check: ok" if not bad else "regress_check: %d figure(s) worse, boxes not comparable -- not failing" % bad)
