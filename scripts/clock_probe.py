#!/usr/bin/env python3
"""Are the scan kernel's launch-time levels the GPU's clock levels?  Runs a command as a CHILD process
and, next to it, samples what the amdgpu driver publishes about the device (no GPU call in this process): the active levels of
pp_dpm_{sclk,mclk,fclk,socclk}, the hwmon power and frequency inputs.  One line per sample on stdout: wall-clock seconds, then
the values.  usage: clock_probe.py <interval seconds> <command ...>   (the command's own timestamps are time.time() too)"""
import glob, os, subprocess, sys, time

def active(path):
    try:
        for ln in open(path).read().splitlines():
            if ln.rstrip().endswith("*"):
                return ln.split(":")[1].strip().rstrip("*").strip()
    except OSError:
        return None
    return "-"

def number(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None

def main():
    dt = float(sys.argv[1])
    devs = sorted(d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(d + "/pp_dpm_sclk"))
    print("# devices:", devs, flush=True)
    if not devs:
        print("# no amdgpu sysfs node is readable here", flush=True)
    child = subprocess.Popen(sys.argv[2:])
    hw = {d: sorted(glob.glob(d + "/hwmon/hwmon*")) for d in devs}
    for d in devs:
        print("#", d, "files:", sorted(os.path.basename(x) for x in glob.glob(d + "/pp_dpm_*")), [sorted(os.listdir(h)) for h in hw[d]], flush=True)
    while child.poll() is None:
        t = time.time()
        for d in devs:
            row = [f"{t:.3f}", os.path.basename(os.path.dirname(d))]
            for c in ("sclk", "mclk", "fclk", "socclk"):
                row.append(f"{c}={active(d + '/pp_dpm_' + c)}")
            for h in hw[d]:
                for f in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"):
                    v = number(h + "/" + f)
                    if v is not None:
                        row.append(f"{f}={v}")
            print(" ".join(row), flush=True)
        time.sleep(dt)
    sys.exit(child.returncode)

main()
