"""One training step as rocprofv3 sees it: from the kernel trace of `bench.py --workload train`, the launches between the last two
adam_kernel launches (= one whole step in steady state, after every autotune probe), summed per kernel name.
usage (GPU box): python scripts/prof_train_step.py <kernel_trace.csv> [top] [step index]   (bench.py: 1 priming step, the warm-up steps, the K timed
steps, K more for the single-stream figure, K more for the roofline region, one step with every kernel timed)"""
import csv, re, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 3:
    sys.exit("fewer than three adam_kernel launches in the trace")
walls = [1e-6 * (int(rows[adam[i + 1]]["End_Timestamp"]) - int(rows[adam[i]]["End_Timestamp"])) for i in range(len(adam) - 1)]
print("step walls (ms) between consecutive adam_kernel launches:", " ".join(f"{w:.2f}" for w in walls))
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(adam) - 2       # the step that ENDS with adam launch `which + 1`
a, b = adam[which], adam[which + 1]
step = rows[a + 1:b + 1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
fam = collections.OrderedDict()
busy = 0
for r in step:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("himo::", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = fam.setdefault(n, [0, 0]); e[0] += 1; e[1] += d
    busy += d
# union of the busy intervals, and the idle gaps between them (who ended before, who started after)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in step)
union, gaps, cur_end, cur_name = 0, [], t0, "adam_kernel"
for a_, b_, n_ in iv:
    if a_ > cur_end:
        gaps.append((a_ - cur_end, cur_name, n_))
        union += b_ - a_
        cur_end, cur_name = b_, n_
    elif b_ > cur_end:
        union += b_ - cur_end
        cur_end, cur_name = b_, n_
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("himo::", ""))[:40]
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} launches, kernel time summed {1e-6 * busy:.3f} ms (streams overlap), "
      f"device busy (union) {1e-6 * union:.3f} ms, idle {1e-6 * sum(g[0] for g in gaps):.3f} ms in {len(gaps)} gaps")
for g, before, after in sorted(gaps, reverse=True)[:12]:
    print(f"   gap {1e-3 * g:7.1f} us  after {short(before)}  before {short(after)}")
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n[:70]:70s} x{c:3d} {1e-3 * t:9.1f} us  avg {1e-3 * t / c:7.1f}")

import os
pat = os.environ.get("LIST")                        # LIST=<substring>: every launch of the matching kernels, in start order
if pat:
    for r in sorted(step, key=lambda r: int(r["Start_Timestamp"])):
        if pat in r["Kernel_Name"]:
            print(f"   {short(r['Kernel_Name'])}  grid {r.get('Grid_Size', '?'):>9s}  wg {r.get('Workgroup_Size', '?'):>5s}  "
                  f"{1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp'])):8.1f} us  stream {r.get('Stream_Id', r.get('Queue_Id', '?'))}")

# per-stream totals: which stream the step's time sits on (the main stream's chain is the critical path when the others hide under it)
per = collections.OrderedDict()
for r in step:
    k = r.get("Stream_Id", r.get("Queue_Id", "?"))
    e = per.setdefault(k, [0, 0, collections.Counter()]); e[0] += 1
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); e[1] += d; e[2][short(r["Kernel_Name"])] += d
for k, (c, t, names) in per.items():
    print(f"stream {k}: {c} launches, {1e-6 * t:.3f} ms of kernel time; top: " + ", ".join(f"{n} {1e-3 * v:.0f} us" for n, v in names.most_common(int(os.environ.get("STREAM_TOP", "6")))))
