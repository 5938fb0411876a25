ulimit -c 0; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rtprof; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for c in $RT_CASES; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$c -o run -- python3 $R/tools/dev/rt_run.py $c random_u32 4 > $OUT/$c.log 2>&1 || exit 1
  tail -1 $OUT/$c.log
  python3 - $OUT/$c <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'vpc_lane' in r['Kernel_Name'] or 'mpc_jit' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
g=(16<<30)/64/64
for k,v in sorted(agg.items()): print(f"   {k:20s} {sum(v)/len(v)/g:10.1f} per group of 64 lines ({len(v)} launches)")
PY
done
