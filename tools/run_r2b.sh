cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o /tmp/xcd_probe 2> gpurun_out/r2b/compile.txt
timeout 300 /tmp/xcd_probe > gpurun_out/r2b/xcd_probe.txt 2>&1
for bs in 1 32; do
  OUT=gpurun_out/r2b/prof_b$bs
  DN_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-latency --batch $bs > $OUT.log 2>&1
  python3 - <<PY
import csv,re,glob
f=glob.glob('$OUT/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
out=open('gpurun_out/r2b/stats_b$bs.txt','w')
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n).replace('void ','')
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3; t=float(r['TotalDurationNs'])/1e3
    if calls < 40: continue
    tot+=t/50
    out.write(f"{n[:60]:60s} calls/step {calls/50:5.1f} avg {avg:7.1f} min {float(r['MinNs'])/1e3:6.1f} us  per-step {t/50:7.1f} us\n")
out.write(f"sum per step {tot:.1f} us\n")
PY
  tail -1 $OUT.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r2b/stats_b$bs.txt
  # keep one step's worth of the trace: start/end per kernel of the last 80 dispatches
  python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/p_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-150:]
t0=int(rows[0]['Start_Timestamp'])
prev=None
with open('gpurun_out/r2b/trace_b$bs.txt','w') as o:
    for r in rows:
        s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
        gap = s-prev if prev is not None else 0
        prev=e
        o.write(f"{s/1e3:9.2f} {e/1e3:9.2f} dur {(e-s)/1e3:7.2f} gap {gap/1e3:6.2f} grid {r.get('Grid_Size','?'):>8} wg {r.get('Workgroup_Size','?'):>5} {r['Kernel_Name'][:70]}\n")
PY
  rm -rf $OUT
done
