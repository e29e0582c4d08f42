#!/bin/bash
# per-form durations of the grouped forward INSIDE a (shortened) bench job: rocprofv3 kernel trace + overlap summary
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_job
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_job -o p -- python3 $R/bench.py --steps 1 --warmup 1 --match-batches 4 --updates 41 --no-cpu-baseline --no-alt-solver --no-phases "$@" > /tmp/prof_job.out 2> /tmp/prof_job.err
tail -2 /tmp/prof_job.err
f=$(find /tmp/prof_job -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
spans = []
for r in rows:
    n = r["Kernel_Name"]
    if "fwd_batch_kernel" in n:
        form = n.split("<")[1].split(">")[0]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        per[form].append((e - s) / 1e3)
        spans.append((s, e, form))
for f, v in sorted(per.items()):
    v = v[len(v) // 2:]
    print("form %s: %d launches, avg %.1f us (second half)" % (f, len(v), sum(v) / len(v)))
# group launches of one update: every fwd_batch kernel up to the update's fwd_loss_kernel (a form may be launched as several
# units -- slices of its items on different lanes -- so "until a form repeats" no longer delimits an update)
ends = sorted(int(r["Start_Timestamp"]) for r in rows if "fwd_loss_kernel" in r["Kernel_Name"])
spans.sort()
groups, cur, k = [], [], 0
for s, e, f in spans:
    while k < len(ends) and ends[k] <= s:
        if cur: groups.append(cur)
        cur = []; k += 1
    cur.append((s, e, f))
if cur: groups.append(cur)
g = groups[len(groups) // 2:]
tot = [(max(e for _, e, _ in x) - min(s for s, _, _ in x)) / 1e3 for x in g]
print("updates %d: wall per update's forward group avg %.1f us; sum of kernel durations avg %.1f us" %
      (len(g), sum(tot) / len(tot), sum(sum((e - s) / 1e3 for s, e, _ in x) for x in g) / len(g)))
x = g[len(g) // 2]; t0 = min(s for s, _, _ in x)
for s, e, f in sorted(x): print("   form %s start +%.1f us dur %.1f us end +%.1f" % (f, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
print("wall of consecutive updates' forward groups (us) and gap to the previous group's end (ms):")
prev = None
for x in groups[-26:]:
    s0, e1 = min(s for s, _, _ in x), max(e for _, e, _ in x)
    print("   %.0f (gap %.2f)" % ((e1 - s0) / 1e3, (s0 - prev) / 1e6 if prev else 0.0), end="")
    prev = e1
print()
others = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "fwd_batch_kernel" not in n:
        others[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(others.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("   other: %-60s calls %5d total %.1f ms avg %.1f us" % (n, len(v), sum(v) / 1e3, sum(v) / len(v)))
PY
