"""Per-kernel medians of one encoder pass under two builds of libgmf_hip.so, alternating in ONE GPU job:
    python tools/lib_kernels_ab.py LIB_A LIB_B [kernel-name filter]"""
import os, sys, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
live = os.path.join(root, "gmf_amd", "libgmf_hip.so")
flt = sys.argv[3] if len(sys.argv) > 3 else "k_"
keep = live + ".keep"; shutil.copy(live, keep)
try:
    for rnd in range(2):
        for lib in sys.argv[1:3]:
            shutil.copy(lib, live)
            subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", "/tmp/lkab", "-o", "run", "--", "python3",
                            os.path.join(root, "tools", "run_scattn_once.py"), "18"], capture_output=True, text=True)
            t = subprocess.run([sys.executable, os.path.join(root, "tools", "time_kernels.py"), "/tmp/lkab/run_kernel_trace.csv"],
                               capture_output=True, text=True).stdout
            print(os.path.basename(lib))
            for l in t.splitlines():
                if flt in l: print("   ", l[:140])
finally:
    shutil.copy(keep, live); os.remove(keep)
