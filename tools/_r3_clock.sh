cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 10 111 174 275 ; do
  rm -rf gpurun_out/clk; 
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk -- python tools/lm_head_ablate.py $v > /dev/null 2>&1
  python - <<PY
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ring_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE":
            rows.append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])) if "End_Timestamp" in r else (float(r["Counter_Value"]),0))
import statistics
if rows:
    c=statistics.median([a for a,b in rows]); d=statistics.median([b for a,b in rows])
    print("PIECES=$v  GRBM_GUI_ACTIVE median", c, "duration ns", d, "=> clock GHz", (c/d if d else None), "keys", None)
else:
    print("no rows", glob.glob("gpurun_out/clk/**/*.csv", recursive=True)[:4])
PY
done
f=$(find gpurun_out/clk -name "*counter_collection.csv" | head -1); head -2 $f
find gpurun_out/clk -name "*.csv" -delete
