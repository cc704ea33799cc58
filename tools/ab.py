"""dev: A/B of environment knobs on one box.   python tools/ab.py "NAME=VAL,NAME2=VAL2" "..." [-- bench args]
Runs bench.py once per setting (no CPU baseline, no latency passes) and prints value / ms / one-at-a-time and the per-family event times."""
import json, os, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for setting in args:
    env = dict(os.environ)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("="); env[k] = v
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-configs", "--steps", "200", "--warmup", "20"] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        print(setting, "FAILED", p.stderr[-600:]); continue
    one = d.get("one_at_a_time", {}).get("ms_per_step")
    lat = d.get("latency", {})
    print(f"[{setting or 'default'}] {d['value']:.0f} img/s  {d['ms_per_step']:.4f} ms in flight  one-at-a-time {one}  sync median {lat.get('median_ms')} list {lat.get('list_api_median_ms')}")
    ks = d.get("kernels", {})
    print("   " + "  ".join(f"{k.split('_kernel')[0]}={v['ms'] * 1e3:.1f}" for k, v in ks.items()))
