"""AMD-SMI utilisation sampler (reference NVML/NVML.cpp:47-88): line format and cadence."""
import os
import re
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "gpu_sampler")
LINE = re.compile(r"^ ?\d{1,2}:\d{1,2}:\d{1,2}:\d{1,3}  Device (\d+): (.+?)  GPU Util: (\d+)  Mem Util: (\d+) Mem Usage: (\d+)$")


@pytest.mark.gpu
def test_sampler_lines_and_cadence():
    subprocess.run(["make", "-C", SRC], check=True, capture_output=True)
    exe = os.path.join(SRC, "amdsmi_sampler")
    t0 = time.time()
    out = subprocess.run([exe, "--count", "7"], check=True, capture_output=True, text=True, timeout=60).stdout
    dt = time.time() - t0
    lines = [l for l in out.split("\n") if l.strip()]
    assert lines, out
    devs = set()
    for l in lines:
        m = LINE.match(l)
        assert m, repr(l)
        devs.add(int(m.group(1)))
        assert 0 <= int(m.group(3)) <= 100 and 0 <= int(m.group(4)) <= 100
    assert len(lines) == 7 * len(devs)
    # 7 ticks at 6 Hz: six full periods between the first and the last tick
    assert 0.9 <= dt <= 3.0, dt


@pytest.mark.gpu
def test_sampler_stops_on_sigint():
    import signal
    subprocess.run(["make", "-C", SRC], check=True, capture_output=True)
    p = subprocess.Popen([os.path.join(SRC, "amdsmi_sampler")], stdout=subprocess.PIPE, text=True)
    time.sleep(0.6)
    p.send_signal(signal.SIGINT)
    out, _ = p.communicate(timeout=20)
    assert p.returncode == 0
    assert sum(1 for l in out.split("\n") if l.strip()) >= 2
