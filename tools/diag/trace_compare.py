import csv,sys
def load(path):
    rows=list(csv.DictReader(open(path)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    idx=[i for i,r in enumerate(rows) if "pair_sym" in r["Kernel_Name"]]
    s=idx[-2]; e=idx[-1]
    return rows[s:e]
def upd(rows):
    return [(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "update_kernel<64, 8, 0>" in r["Kernel_Name"]]
cols=[upd(load("gpurun_out/r2_tr_%s/t_kernel_trace.csv"%t)) for t in sys.argv[1:]]
print("sums", " ".join("%s=%.0f"%(t,sum(c)) for t,c in zip(sys.argv[1:],cols)))
for i in range(0,len(cols[0]),3):
    print(i, " ".join("%7.1f (%.3f)"%(c[i],c[i]/cols[0][i]) for c in cols))
