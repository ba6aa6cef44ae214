"""Reduce rocprofv3 --pmc counter_collection CSVs to HBM bytes per launch per kernel family.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per MI355X_MICROARCH.md section HBM, on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so the read side is doubled; WRITE_SIZE is taken
as reported (uncalibrated)."""
import csv, json, re, sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
FAMILIES = {"compdis_kernel": ("compdis_kernel<",), "frame_prep_kernel": ("frame_prep_kernel",),
            "conv3x3_mfma_kernel": ("conv_mfma_kernel<3, 1,",), "conv3x3s2_mfma_kernel": ("conv_mfma_kernel<3, 2,",),
            "conv1x1_mfma_kernel": ("conv_mfma_kernel<1, 1,",), "pillar_feature_kernel": ("pillar_feature_kernel",),
            # the 20 stride-1 3x3 layers of a forward in split precision: every kernel structure (autotuned per layer; the
            # last template argument of the weights-from-L2 kernels is the stride)
            # conv3_presplit_kernel<EPI, PH, MI, OSPLIT, S, NT>: the stride is the FIFTH argument
            "conv3x3_split_kernel": (r"conv3_split_kernel<.*, 1>", "conv_bf16x3_kernel<3,", r"conv3_presplit_kernel<\d+, \d+, \d+, \w+, 1, \d+>"),
            "conv3x3s2_split_kernel": (r"conv3_split_kernel<.*, 2>", r"conv3_presplit_kernel<\d+, \d+, \d+, \w+, 2, \d+>"),
            "conv1x1_split_kernel": ("conv_bf16x3_kernel<1,", "conv1_presplit_kernel<"), "gru_head_kernel": ("gru_head_kernel",),
            "upsample2x_kernel": ("upsample2x_kernel",),
            "nsf_forward_kernel": ("nsf_forward_kernel",), "nsf_backward_kernel": ("nsf_backward_kernel",), "nsf_update_kernel": ("nsf_update_kernel",)}
# each kernel family is read from the workload whose bench configuration is the quoted one
SOURCE = {"compdis_kernel": "compdis", "frame_prep_kernel": "compdis", "nsf_forward_kernel": "fastnsf", "nsf_backward_kernel": "fastnsf",
          "nsf_update_kernel": "fastnsf"}
# Only the launches of the LAST timed step count: the run starts with the network's one-off tile autotune, which launches every
# variant of every layer (slower tiles, more halo traffic) and would otherwise be averaged into the tuned kernels' figure.
# bench.py runs: priming pass, warm-up, parity pass, K timed steps, one fully profiled step -- all with identical launch lists
# once tuned, so "the last 1 / (K + 3)" of a family's launches in dispatch order is one clean step.
KEEP_LAST_FRACTION = {"pipeline": 0.2, "compdis": 1.0, "fastnsf": 0.5}
# launches per bench step (16 samples per step) of the pipeline's kernel families: exactly the last TWO steps are averaged, so
# every layer of a family enters with equal weight
PER_STEP = {"conv3x3_split_kernel": 20, "conv3x3s2_split_kernel": 3, "conv1x1_split_kernel": 6, "upsample2x_kernel": 3,
            "pillar_feature_kernel": 4, "gru_head_kernel": 1, "conv3x3_mfma_kernel": 20, "conv3x3s2_mfma_kernel": 3, "conv1x1_mfma_kernel": 6}
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sorted(root.glob("*_*_SIZE")):
    counter = "FETCH_SIZE" if d.name.endswith("FETCH_SIZE") else "WRITE_SIZE"
    workload = d.name.split("_")[0]
    per_family = defaultdict(list)
    for f in d.rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                for fam, pats in FAMILIES.items():
                    if any(re.search(pat, name) for pat in pats) and SOURCE.get(fam, "pipeline") == workload:
                        per_family[fam].append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
    for fam, rows in per_family.items():
        rows.sort()
        if workload == "pipeline" and fam in PER_STEP and len(rows) >= 2 * PER_STEP[fam]:
            keep = rows[-2 * PER_STEP[fam]:]
        else:
            keep = rows[-max(1, int(len(rows) * KEEP_LAST_FRACTION.get(workload, 1.0))):]
        a = acc[fam][counter]
        a[0] += sum(v for _, v in keep); a[1] += len(keep)
out = {}
for fam, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and c["FETCH_SIZE"][1] and c["WRITE_SIZE"][1]:
        rd = c["FETCH_SIZE"][0] / c["FETCH_SIZE"][1] * 1024 * 2      # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        wr = c["WRITE_SIZE"][0] / c["WRITE_SIZE"][1] * 1024
        out[fam] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch_x2_corrected": rd, "write_bytes_per_launch": wr,
                    "launches_sampled": c["FETCH_SIZE"][1]}
print(json.dumps(out, indent=1))
