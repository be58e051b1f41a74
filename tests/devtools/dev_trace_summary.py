"""Summary of a rocprofv3 --kernel-trace CSV: per kernel calls / total / mean, and for the steady state (the last `--iters` iterations,
found by a marker kernel that runs once per iteration) the GPU-busy fraction and the idle gaps between consecutive kernels.
    python tests/devtools/dev_trace_summary.py <kernel_trace.csv> [--marker adam_step] [--iters 10] > profiles/rNN_..._kernel_stats.md"""
import argparse, collections, csv, re, sys

def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.replace("gof::", "").replace("at::native::", "")[:70]

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--marker", default="adam_step")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
if len(marks) > a.iters:
    lo, hi = marks[-a.iters - 1] + 1, marks[-1] + 1
    n_it = a.iters
else:
    lo, hi, n_it = 0, len(rows), max(1, len(marks))
win = rows[lo:hi]
span = win[-1][1] - win[0][0]
busy = sum(e - s for s, e, _ in win)
per = collections.OrderedDict()
for s, e, k in win:
    d = per.setdefault(k, [0, 0]); d[0] += 1; d[1] += e - s
gaps = []
for (s0, e0, k0), (s1, e1, k1) in zip(win, win[1:]):
    gaps.append((max(0, s1 - e0), k0, k1))
print("# kernel trace summary: %s\n" % a.csv.split("/")[-1])
print("steady state = the last %d iterations (marker kernel `%s`): %d kernel launches, span %.3f ms per iteration, GPU busy %.3f ms per iteration (%.1f %%), idle %.3f ms per iteration\n"
      % (n_it, a.marker, len(win), span / n_it / 1e6, busy / n_it / 1e6, 100.0 * busy / span, (span - busy) / n_it / 1e6))
print("| kernel | launches / iteration | us / iteration | mean us |\n|---|---|---|---|")
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %.1f | %.1f | %.1f |" % (k, c / n_it, t / n_it / 1e3, t / c / 1e3))
print("\nlargest idle gaps (us, after kernel -> before kernel), per iteration sums by pair:\n")
pair = collections.Counter()
for g, k0, k1 in gaps:
    pair[(k0, k1)] += g
print("| after | before | idle us / iteration |\n|---|---|---|")
for (k0, k1), g in pair.most_common(25):
    print("| %s | %s | %.1f |" % (k0, k1, g / n_it / 1e3))
