import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/run_counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "encode_placed" in n or "encode_frames" in n or "compact_frames" in n:
        agg[n[:40]][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(n[:40],r["Counter_Name"])]+=1
for n in agg:
    print(n, {k: round(v/cnt[(n,k)],1) for k,v in agg[n].items()}, "dispatches", max(cnt[(n,k)] for k in agg[n]))
