#!/usr/bin/env python3
"""Distribution of ONE kernel's duration in a rocprofv3 kernel trace (mean, deciles over the run, quantiles):
tools/kernel_dist.py <rocprofv3 output dir> <kernel name fragment>"""
import csv,glob,sys
import numpy as np
f=glob.glob(sys.argv[1]+"/runc/*kernel_trace.csv")[0]
d=[]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
d=np.array([x[1] for x in sorted(d)])/1e3
print(sys.argv[2], "n", len(d), "mean", round(float(d.mean()),2), "deciles", [round(float(x.mean()),1) for x in np.array_split(d,10)])
print("quantiles",[round(float(np.quantile(d,q)),1) for q in (0.05,0.25,0.5,0.75,0.95,0.99)])
