"""Run ON THE GPU BOX: socket power / shader clock (hwmon sysfs of the first amdgpu card) sampled every 20 ms while a child
command runs; prints the cap, and mean / p95 / max of power and clock over the child's lifetime minus its first seconds.
usage: power_trace.py <skip_seconds> <command...>        (the sampler itself never touches the GPU)"""
import glob, subprocess, sys, time
skip = float(sys.argv[1])
cmd = sys.argv[2:]
hws = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if glob.glob(h + "/power1_*")]
if not hws:
    print("no amdgpu hwmon with power1_* found:", glob.glob("/sys/class/drm/card*/device/hwmon/*"))
    sys.exit(subprocess.call(cmd))
def rd(hw, name):
    try:
        return float(open(f"{hw}/{name}").read())
    except Exception:
        return float("nan")
pname = "power1_average" if glob.glob(hws[0] + "/power1_average") else "power1_input"
p = subprocess.Popen(cmd)
t0 = time.time()
allrows = {h: [] for h in hws}
while p.poll() is None:
    t = time.time() - t0
    for h in hws:       # the box shows all cards of the host in sysfs; the one this job runs on is the one that draws power
        allrows[h].append((t, rd(h, pname) / 1e6, rd(h, "freq1_input") / 1e6, rd(h, "temp1_input") / 1e3))
    time.sleep(0.02)
mean_p = {h: sum(r[1] for r in v if r[0] >= skip and r[1] == r[1]) / max(1, sum(1 for r in v if r[0] >= skip)) for h, v in allrows.items()}
hw = max(mean_p, key=mean_p.get)
cap = rd(hw, "power1_cap") / 1e6
print("mean power per card (W):", " ".join(f"{mean_p[h]:.0f}" for h in hws))
rows = [r for r in allrows[hw] if r[0] >= skip]
import statistics as st
def summ(i):
    v = sorted(r[i] for r in rows if r[i] == r[i])
    if not v:
        return "n/a"
    return f"mean {st.mean(v):.0f}  p50 {v[len(v) // 2]:.0f}  p95 {v[int(len(v) * 0.95)]:.0f}  max {v[-1]:.0f}"
print(f"hwmon {hw}: cap {cap:.0f} W; {len(rows)} samples after the first {skip:.0f} s")
print("power W   :", summ(1))
print("sclk MHz  :", summ(2))
print("temp C    :", summ(3))
sys.exit(p.returncode)
