import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
key=sys.argv[2] if len(sys.argv)>2 else "k_grid_frame"
def nm(r):
    m=re.search(r"(k_\w+(<[^>]*>)?|__amd\w+|at::native::\w+|rocprim\S{0,40})", r["Kernel_Name"]); return (m.group(1) if m else r["Kernel_Name"])[:52]
starts=[i for i,r in enumerate(rows) if key in r["Kernel_Name"]]
spans=[(int(rows[starts[j+1]]["Start_Timestamp"])-int(rows[starts[j]]["Start_Timestamp"]))/1e3 for j in range(len(starts)-1)]
import statistics
med=statistics.median(spans)
j=min(range(len(spans)), key=lambda j:abs(spans[j]-med))
i0,i1=starts[j],starts[j+1]
t0=int(rows[i0]["Start_Timestamp"]); prev=t0
for r in rows[i0:i1]:
    s=int(r["Start_Timestamp"]);e=int(r["End_Timestamp"])
    print(f"{nm(r):52s} {(e-s)/1e3:7.1f} us  at {(s-t0)/1e3:8.1f}  gap {(s-prev)/1e3:6.1f} q{r.get('Queue_Id','')} grid {r.get('Grid_Size','')}/{r.get('Workgroup_Size','')}")
    prev=max(prev,e)
print('median span',med,'this',spans[j], 'launches', i1-i0)
