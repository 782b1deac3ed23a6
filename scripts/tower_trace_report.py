import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
starts = [i for i, n in enumerate(names) if "embed_ln" in n]
i0 = starts[-1]
seq = list(zip(names[i0:], dur[i0:]))
tot = sum(d for _, d in seq)
print("last tower pass: %d launches, %.2f ms of kernel time, wall %.2f ms" % (len(seq), tot / 1e3, (int(rows[-1]["End_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e6))
agg = collections.OrderedDict()
for n, d in seq:
    k = n[:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("  %-70s %4d  %8.1f us avg  %7.2f ms" % (k, c, t / c, t / 1e3))
print("block 12, in launch order:")
blk = [i for i, (n, _) in enumerate(seq) if "time_attn" in n]
if len(blk) > 12:
    for n, d in seq[blk[11] - 1:blk[12] - 1]:
        print("     %-80s %8.1f us" % (n[:80], d))
