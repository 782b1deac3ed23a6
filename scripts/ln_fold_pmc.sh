#!/bin/bash
# HBM-side traffic of ONE vision-tower pass with and without the LayerNorm fold (rocprofv3 --pmc, one counter per pass) -> gpurun_out/ln_fold_pmc/ln_fold_pmc.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ln_fold_pmc; mkdir -p $O
for v in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_${v}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_${v}_$c -- python3 $R/scripts/tower_trace.py $v > $O/pmc_${v}_$c.log 2>&1
done; done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
def load(v,c):
    f=glob.glob("/tmp/pm_%s_%s/**/*counter_collection.csv"%(v,c),recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if r["Counter_Name"]==c]
    rows.sort(key=lambda r:int(r["Dispatch_Id"]))
    # last tower pass: from the last embed_ln launch on
    names=[r["Kernel_Name"] for r in rows]
    i0=max(i for i,n in enumerate(names) if "embed_ln" in n)
    agg=collections.OrderedDict()
    for r in rows[i0:]:
        a=agg.setdefault(r["Kernel_Name"][:64],[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
    return agg
M=32*4097; D=1024; EL=M*D
out=["# r4: HBM-side traffic of one vision-tower pass (config 2, B = 32), LayerNorm fold on / off","",
"`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) `-- python3 scripts/tower_trace.py 1|0`; last of the 11 passes; FETCH_SIZE x 2 (gfx950 correction), KiB -> bytes.",
"Bytes per element and block = bytes / (M x D = %d x %d elements) / 24 blocks."%(M,D),""]
tot={}
for v,name in ((1,"fold"),(0,"no fold")):
    f=load(v,"FETCH_SIZE"); w=load(v,"WRITE_SIZE")
    out+=["## "+name,"","| kernel | launches | fetch MiB / launch | write MiB / launch | total GiB / pass | B / element / block |","|---|---|---|---|---|---|"]
    T=0.0
    for k,(n,fv) in f.items():
        wv=w.get(k,[n,0.0])[1]
        fb=2*fv*1024; wb=wv*1024
        T+=fb+wb
        if (fb+wb)/2**30>0.05:
            out.append("| %s | %d | %.1f | %.1f | %.2f | %.2f |"%(k.replace("|","/"),n,fb/n/2**20,wb/n/2**20,(fb+wb)/2**30,(fb+wb)/EL/24))
    out+=["","total: %.1f GiB per pass = %.1f B per element and block"%(T/2**30,T/EL/24),""]
    tot[name]=T
out+=["Difference: %.1f GiB per pass = %.1f B per element and block removed by the fold."%((tot["no fold"]-tot["fold"])/2**30,(tot["no fold"]-tot["fold"])/EL/24),""]
open(R+"/gpurun_out/ln_fold_pmc/ln_fold_pmc.md","w").write("\n".join(out))
print("\n".join(out))
PY
