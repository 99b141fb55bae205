cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/gllc; rm -rf $O; mkdir -p $O
i=0
for group in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/pass$i -o p -- python3 tools/bench_gll.py > $O/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set); dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/gllc/pass*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:44]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); calls[k].add((f,r["Dispatch_Id"]))
for f in glob.glob('gpurun_out/gllc/pass1/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:44]
        dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6)
for k,v in sorted(agg.items(), key=lambda kv:-sum(dur[kv[0]])):
    if 'gll' in k:
        n=len(calls[k])/2; print(k, "dispatches", n, "total ms %.2f"%sum(dur[k]), {a:round(b/max(n,1)) for a,b in sorted(v.items())})
PY
