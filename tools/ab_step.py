"""A/B of builds / knob settings of libgmf_hip.so in ONE GPU job (box-to-box spread on the pool is +-5 %): every arm is a child
process running tools/kernel_times.py (whole forward, torch-profiler kernel table), arms alternating, two rounds.
    python tools/ab_step.py "LIB[:knob=v,knob=v]" "LIB[:...]" ...  [--filter k_linear,k_scattn] [--B 32] [--N 5000]
LIB: a path, or `live` for gmf_amd/libgmf_hip.so."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arms, flt, B, N = [], ["k_linear", "k_scattn", "encode"], "32", "5000"
it = iter(sys.argv[1:])
for a in it:
    if a == "--filter":
        flt = next(it).split(",") + ["encode"]
    elif a == "--B":
        B = next(it)
    elif a == "--N":
        N = next(it)
    else:
        arms.append(a)
for rnd in range(2):
    for arm in arms:
        lib, _, knobs = arm.partition(":")
        env = dict(os.environ, FULL="1", ROWS="14")
        if lib != "live":
            env["GMF_LIB"] = lib if os.path.isabs(lib) else os.path.join(root, lib)
        cmd = [sys.executable, os.path.join(root, "tools", "kernel_times.py"), B, N, "8"] + [k for k in knobs.split(",") if k]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env)
        print(f"[round {rnd}] {arm}")
        lines = [l for l in out.stdout.splitlines() if any(f in l for f in flt)]
        for l in lines:
            print("    " + l[:150])
        if not lines:
            print("    (no output) " + out.stderr[-400:])
        sys.stdout.flush()
