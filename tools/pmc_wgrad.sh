cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/wgrad_bench.py --blocks 512,1024,1536,2048 2>&1 | tail -9
for c in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/wp_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/wp_$n -o w -- python3 tools/wgrad_bench.py --blocks 1024 --only l3.c2 --iters 2 > gpurun_out/wp_$n.log 2>&1
  tail -1 gpurun_out/wp_$n.log | cut -c1-200
  python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/wp_$n/*counter_collection.csv')
if f:
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'wgrad_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
done
