cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/scripts/micro/ldlt_mfma_time.bin
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_F64"; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm --output-format csv -- $B > /tmp/pm.log 2>&1
  f=$(ls /tmp/pm/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $set ($f)"
  python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# dispatches in order; group per kernel name+dispatch id
d=collections.OrderedDict()
for r in rows:
    k=(int(r['Dispatch_Id']),r['Kernel_Name'][:28])
    d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
for i,(k,v) in enumerate(d.items()):
    if i in (5,13,23):  # stride 21/256, packed 32/256, packed 32/512
        print(k, {a:int(b) for a,b in v.items()})
PY
done
tail -3 /tmp/pm.log
