set -o pipefail
D=gpurun_out/r3d; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 $GRAFT_REPO_ROOT/bench/diag/onek_ablate.py --child --batch 1024 > $GRAFT_REPO_ROOT/$D/pmc1.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
python3 - "$f" > $D/pmc1_summary.txt <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=(r['Kernel_Name'][:70], r.get('Grid_Size',''))
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='TCC_EA0_WRREQ_sum': cnt[k]+=1
for k,v in agg.items():
    n=max(cnt[k],1)
    print(k, 'calls',n, {c: round(x/n) for c,x in v.items()})
PY
cat $D/pmc1_summary.txt | head -30
