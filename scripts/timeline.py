#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV between the last two marker launches (k_madd_rate, scripts/pipeline_run.py):
how long the GPU ran kernels of one / two / three queues at once, per-kernel durations, the idle gaps, and an excerpt
of the dispatch sequence with the queue of every kernel.
    python3 scripts/timeline.py <kernel_trace.csv> [marker substring] [excerpt lines]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_madd_rate"
excerpt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["name"] = r["Kernel_Name"].split("(")[0][:40]
    r["q"] = r.get("Queue_Id") or r.get("Stream_Id") or "?"
rows.sort(key=lambda r: r["s"])
marks = [r for r in rows if marker in r["Kernel_Name"]]
# markers come in pairs (warm-up launch + timed launch of vmpc_ed25519_madd_rate): region = between the end of the
# last-but-one group and the start of the last group
groups = []
for m in marks:
    if groups and m["s"] - groups[-1][-1]["e"] < 2_000_000:
        groups[-1].append(m)
    else:
        groups.append([m])
assert len(groups) >= 2, f"need two marker groups, found {len(groups)}"
t_lo, t_hi = groups[-2][-1]["e"], groups[-1][0]["s"]
reg = [r for r in rows if r["s"] >= t_lo and r["e"] <= t_hi]
span = (t_hi - t_lo) / 1e3
print(f"region {span:.1f} us, {len(reg)} kernels, queues {sorted(set(r['q'] for r in reg))}")
# concurrency profile: sweep
ev = []
for r in reg:
    ev.append((r["s"], 1, r["q"]))
    ev.append((r["e"], -1, r["q"]))
ev.sort()
active = defaultdict(int)
last = t_lo
time_at = defaultdict(float)       # number of queues with a kernel in flight -> us
for t, d, q in ev:
    k = sum(1 for v in active.values() if v > 0)
    time_at[k] += (t - last) / 1e3
    last = t
    active[q] += d
time_at[0] += (t_hi - last) / 1e3
for k in sorted(time_at):
    print(f"  {k} queue(s) busy: {time_at[k]:9.1f} us  {100 * time_at[k] / span:5.1f} %")
# per-kernel stats
st = defaultdict(lambda: [0, 0.0])
for r in reg:
    st[r["name"]][0] += 1
    st[r["name"]][1] += (r["e"] - r["s"]) / 1e3
print("kernel                                    launches   total us    avg us   share of span")
for name, (c, tot) in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:40s} {c:6d} {tot:10.1f} {tot / c:9.1f} {100 * tot / span:8.1f} %")
# overlap of the dominant kernel with other queues' kernels
dom = max(st, key=lambda k: st[k][1])
dom_rows = [r for r in reg if r["name"] == dom]
others = [r for r in reg if r["name"] != dom]
ov = defaultdict(float)
for d in dom_rows:
    for o in others:
        if o["q"] == d["q"] or o["e"] <= d["s"] or o["s"] >= d["e"]:
            continue
        ov[o["name"]] += (min(o["e"], d["e"]) - max(o["s"], d["s"])) / 1e3
print(f"time other queues' kernels spent in flight WHILE a {dom} of another queue was in flight:")
for name, us in sorted(ov.items(), key=lambda kv: -kv[1]):
    print(f"  {name:40s} {us:10.1f} us of its {st[name][1]:10.1f} us")
print(f"dispatch excerpt (first {excerpt} kernels of the region): start us, duration us, queue, kernel")
for r in reg[:excerpt]:
    print(f"  {(r['s'] - t_lo) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r['q']:>3s}  {r['name']}")
