import csv,glob,collections,sys
d=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set)
for f in glob.glob(d+"/*/*counter_collection.csv")+glob.glob(d+"/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:28]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    if any(x in k for x in sys.argv[2:]):
        n=len(calls[k]); print(k, "dispatches", n, {a:round(b/n) for a,b in sorted(v.items())})
