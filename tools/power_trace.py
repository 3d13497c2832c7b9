#!/usr/bin/env python3
"""Shader clock and socket power sampled at ~100 Hz from the amdgpu hwmon files while a command keeps the GPU busy (GPU box):

    python tools/power_trace.py <label> -- <command ...>

Prints one summary line: samples, and mean / p10 / p50 / p90 of sclk (MHz) and power (W) over the BUSY window (samples whose power is above
idle + 40 % of the idle-to-peak span) -- the evidence behind "the launch sits at the power budget" (DESIGN.md): a kernel pair whose clocks per
step fall while the chip's clock falls with them shows here as a lower sclk at the same power."""
import glob
import subprocess
import sys
import time


def find(patterns):
    for p in patterns:
        hits = sorted(glob.glob(p))
        if hits:
            return hits[0]
    return None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().split()[0])
    except Exception:      # noqa: BLE001
        return None


def main():
    label = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    # every amdgpu hwmon of the box is sampled (a node shows all of its GPUs in sysfs, whichever one the process may use); the card whose
    # power moves most over the run is the one the command ran on
    cards = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f_pow = find([hw + "/power1_average", hw + "/power1_input"])
        f_clk = find([hw + "/freq1_input"])
        if f_pow and f_clk:
            cards.append((hw.split("/")[4], f_pow, f_clk))
    if not cards:
        print(f"{label}: no hwmon power / clock files")
        return subprocess.call(cmd)
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    allrows = {c[0]: [] for c in cards}
    t0 = time.perf_counter()
    while p.poll() is None:
        for name, f_pow, f_clk in cards:
            w, hz = read_int(f_pow), read_int(f_clk)
            if w is not None and hz is not None:
                allrows[name].append((time.perf_counter() - t0, w / 1e6, hz / 1e6))
        time.sleep(0.004)
    span = {n: (max(r[1] for r in v) - min(r[1] for r in v)) if v else 0.0 for n, v in allrows.items()}
    pick = max(span, key=span.get)
    rows = allrows[pick]
    f_clk, f_pow = next((c[2], c[1]) for c in cards if c[0] == pick)
    if len(rows) < 20:
        print(f"{label}: only {len(rows)} samples")
        return p.returncode
    pw = sorted(r[1] for r in rows)
    idle, peak = pw[len(pw) // 20], pw[-1]
    busy = [r for r in rows if r[1] > idle + 0.4 * (peak - idle)]
    if len(busy) < 5:
        busy = rows

    def q(v, f):
        v = sorted(v)
        return v[min(len(v) - 1, int(f * len(v)))]
    clk, po = [r[2] for r in busy], [r[1] for r in busy]
    rate = len(rows) / max(rows[-1][0], 1e-9)
    print(f"{label}: {len(rows)} samples at {rate:.0f} Hz, {len(busy)} busy | sclk MHz mean {sum(clk) / len(clk):.0f} p10 {q(clk, .1):.0f} p50 {q(clk, .5):.0f} "
          f"p90 {q(clk, .9):.0f} | power W mean {sum(po) / len(po):.0f} p10 {q(po, .1):.0f} p50 {q(po, .5):.0f} p90 {q(po, .9):.0f} | idle {idle:.0f} W peak {peak:.0f} W "
          f"| {pick} of {len(cards)} cards ({f_clk.split('/')[-1]}, {f_pow.split('/')[-1]})")
    return p.returncode


if __name__ == "__main__":
    sys.exit(main())
