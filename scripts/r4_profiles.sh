#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/make_profiles.sh r4 > gpurun_out/r4_make_profiles.log 2>&1
bash scripts/make_profiles.sh r4_c4 --config c4 --batch 4 > gpurun_out/r4_c4_make_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_new
rm -rf /tmp/sv_sq /tmp/sv_fetch /tmp/sv_write /tmp/sv_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sv_kt -- python3 $R/scripts/space_variants.py 1 2 3 > $OUT/r4_space_variants_kt.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/sv_sq -- python3 $R/scripts/space_variants.py 1 2 3 > $OUT/r4_space_variants_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sv_fetch -- python3 $R/scripts/space_variants.py 1 2 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sv_write -- python3 $R/scripts/space_variants.py 1 2 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]; OUT = R + "/gpurun_out/profiles_new/"
L = ["# r4 -- space attention variants at the headline shape (B = 32, T = 16, n = 256, 16 heads, head-major planes), 12 calls each", "",
     "`python3 scripts/space_variants.py 1 2 3` under rocprofv3: --kernel-trace --stats (durations), --pmc SQ_* (one pass), --pmc FETCH_SIZE / WRITE_SIZE (separate passes).",
     "MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); VALU busy % = 4 x SQ_ACTIVE_INST_VALU / (4 x SQ_BUSY_CU_CYCLES); wait % = SQ_WAIT_ANY / SQ_WAVE_CYCLES;",
     "traffic = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE per launch; algorithmic bytes per launch: 8 * B * N * D = 1074 MB.", "",
     "| kernel | avg us (kernel-trace) | MFMA busy % | VALU busy % | wave wait % | traffic MiB / launch |", "|---|---|---|---|---|---|"]
dur = {}
for f in glob.glob("/tmp/sv_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "space_attn" in r["Name"]: dur[r["Name"]] = float(r["AverageNs"]) / 1e3
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/sv_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "space_attn" in r["Kernel_Name"]: sq[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
tr = collections.defaultdict(lambda: [0.0, 0.0, 0])
for c, d, i in (("FETCH_SIZE", "/tmp/sv_fetch", 0), ("WRITE_SIZE", "/tmp/sv_write", 1)):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "space_attn" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tr[r["Kernel_Name"]][i] += float(r["Counter_Value"])
                if i == 0: tr[r["Kernel_Name"]][2] += 1
for k in sorted(dur):
    c = sq.get(k, {})
    busy = 4.0 * c.get("SQ_BUSY_CU_CYCLES", 0) or 1.0
    t = tr.get(k, [0, 0, 1]); n = max(t[2], 1)
    L.append("| %s | %.1f | %.1f | %.1f | %.1f | %.1f |" % (k[:70].replace("|", "/"), dur[k], 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy, 100 * 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / busy,
             100 * c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), (2 * t[0] + t[1]) / n / 1024.0))
open(OUT + "r4_space_variants.md", "w").write("\n".join(L) + "\n")
print("\n".join(L[-5:]))
PY
cd $R
python scripts/space_probe.py > $OUT/r4_space_probe.txt 2>&1
ls $OUT
