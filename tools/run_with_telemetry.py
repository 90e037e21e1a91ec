"""Run a command while sampling socket power / shader clock (neo360_amd.telemetry, librocm_smi64), print the command's
output and the means over the BUSY samples (power >= 80 % of the highest sample: host-side set-up is excluded).
usage: python tools/run_with_telemetry.py <command> [args...]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo360_amd import telemetry

with telemetry.Sampler(0, period_s=0.02) as s:
    rc = subprocess.call(sys.argv[1:])
pw, ck = s.power, s.sclk
if pw:
    top = max(pw)
    busy = [i for i, p in enumerate(pw) if p >= 0.8 * top]
    mean = lambda xs: sum(xs) / max(len(xs), 1)
    print("telemetry: %d samples, %d busy; busy power %.0f W (max %.0f, limit %s), busy sclk %.0f MHz"
          % (len(pw), len(busy), mean([pw[i] for i in busy]), top, s.cap, mean([ck[i] for i in busy if i < len(ck)])))
else:
    print("telemetry:", s.summary().get("telemetry"))
sys.exit(rc)
