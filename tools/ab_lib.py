"""A/B of two builds of libgmf_hip.so in ONE GPU job (device-to-device spread on the pool is +-5 %): runs
tests/tools/ab_scattn.py in a child process per library, alternating.   python tools/ab_lib.py LIB_A LIB_B [ab_scattn args...]"""
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
args = sys.argv[3:] or ["32", "5000", "18", "compat_cache=1"]
live = os.path.join(root, "gmf_amd", "libgmf_hip.so")
keep = live + ".keep"
shutil.copy(live, keep)
try:
    for rnd in range(2):
        for lib in libs:
            shutil.copy(lib, live)
            out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "ab_scattn.py")] + args, capture_output=True, text=True)
            print(os.path.basename(lib), "|", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
finally:
    shutil.copy(keep, live)
    os.remove(keep)
