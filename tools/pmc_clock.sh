#!/bin/bash
# Dev tool (GPU box): effective shader clock during ONE conv layer/config = GRBM_GUI_ACTIVE cycles / kernel duration.
# usage: tools/pmc_clock.sh OUTDIR batch c h w n size stride pad act cfg
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; shift
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/clk -- python3 $R/tools/conv_one.py "$@" 40 > $R/$OUT/clk.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
d="$R/$OUT/clk"
dur={}
for r in csv.DictReader(open(glob.glob(d+"/*/*kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
rows=[(float(r["Counter_Value"]),dur[r["Dispatch_Id"]]) for r in csv.DictReader(open(glob.glob(d+"/*/*counter_collection.csv")[0])) if "conv" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE"]
rows=rows[5:]
import statistics
mhz=[c/ns*1e3 for c,ns in rows]
print("launches %d  median duration %.1f us  effective clock: median %.0f MHz  min %.0f  max %.0f" % (len(rows), statistics.median(ns for _,ns in rows)/1e3, statistics.median(mhz), min(mhz), max(mhz)))
PY
