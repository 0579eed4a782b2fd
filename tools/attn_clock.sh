cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pc1 -- python $GRAFT_REPO_ROOT/tools/attn_fwd_time.py > /tmp/pc1.log 2>&1
f=$(find /tmp/pc1 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if 'attn_fwd' in r['Kernel_Name']:
        acc[r['Counter_Name']].append((float(r['Counter_Value']), float(r['End_Timestamp'])-float(r['Start_Timestamp'])))
for k,v in acc.items():
    v=v[20:]
    print(k, 'avg', sum(x for x,_ in v)/len(v), 'avg_ns', sum(t for _,t in v)/len(v), 'n', len(v))
PY
