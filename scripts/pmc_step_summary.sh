#!/bin/bash
# One table for a whole pipeline bench step: per kernel family the launches per step, time, effective shader clock
# (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles) and the fabric-side
# bytes (FETCH_SIZE x 2 per the gfx950 note, WRITE_SIZE) -> achieved GB/s.  Three rocprofv3 --pmc passes over
# `bench.py --no-cpu-baseline --no-extra-precisions --no-extra-workloads --steps 2 --warmup 1`; only the launches of the last two steps are averaged.
# usage (GPU box, repo root): bash scripts/pmc_step_summary.sh > gpurun_out/pmc_step_summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcstep
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $R/bench.py --no-cpu-baseline --no-extra-precisions --no-extra-workloads --steps 2 --warmup 1 > $OUT/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, re, collections
FAM = [("conv3x3 (stride 1)", r"conv3_presplit_kernel<\d+, \d+, \d+, \w+, 1, \d+>", 20), ("conv3x3 (stride 2)", r"conv3_presplit_kernel<\d+, \d+, \d+, \w+, 2, \d+>", 3),
       ("conv1x1", r"conv1_presplit_kernel<", 4), ("gru_head", r"gru_head_kernel<", 1), ("upsample2x", r"upsample2x_kernel<", 3),
       ("pillar_feature", r"pillar_feature_kernel", 4), ("pillar_fill", r"pillar_fill_kernel", 4), ("pillar_assign", r"pillar_assign_kernel", 4),
       ("compdis", r"compdis_kernel<", 1)]
vals = collections.defaultdict(lambda: collections.defaultdict(list))      # family -> counter -> [(dispatch, value, ns)]
for p in ("p1", "p2", "p3"):
    dur = {}
    for f in glob.glob(f"$OUT/{p}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(f"$OUT/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for name, pat, per in FAM:
                if re.search(pat, r["Kernel_Name"]):
                    vals[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0)))
print(f"{'kernel family':20s} {'per step':>8s} {'us/launch':>10s} {'ms/step':>8s} {'clock GHz':>9s} {'MFMA busy':>9s} {'read MB':>9s} {'write MB':>9s} {'GB/s':>7s}")
tot = 0.0
for name, pat, per in FAM:
    c = vals.get(name)
    if not c or "GRBM_GUI_ACTIVE" not in c: continue
    last = lambda k: sorted(c[k])[-2 * per:] if k in c else []
    g = last("GRBM_GUI_ACTIVE"); m = last("SQ_VALU_MFMA_BUSY_CYCLES"); fr = last("FETCH_SIZE"); wr = last("WRITE_SIZE")
    ns = sum(x[2] for x in g) / len(g)
    gui = sum(x[1] for x in g) / len(g) / 8
    busy = (sum(x[1] for x in m) / len(m) / 1024) / gui if m and gui else float("nan")
    rd = sum(x[1] for x in fr) / len(fr) * 1024 * 2 / 1e6 if fr else float("nan")
    wb = sum(x[1] for x in wr) / len(wr) * 1024 / 1e6 if wr else float("nan")
    tot += ns * per / 1e6
    print(f"{name:20s} {per:8d} {ns/1e3:10.1f} {ns*per/1e6:8.3f} {gui/ns:9.2f} {busy:9.1%} {rd:9.1f} {wb:9.1f} {(rd + wb) * 1e6 / ns if ns else 0:7.0f}")
print(f"sum {tot:.2f} ms per 16-sample step (kernels listed)")
PY
