"""Timing-only ablations of the attention kernel k_scattn_h2p (results WRONG by construction; never part of the library).
Each ablation is a textual patch applied to a scratch COPY of gmf_amd/csrc/encoder_kernels.hip; the patched object is
linked with the product's other objects into tools/_ab/libgmf_hip_<name>.so (git-ignored, travels with gpurun).

    python tools/ubench/ablate_h2p.py build            # here (cross-compiles)
    python tools/ubench/ablate_h2p.py run [B] [N]      # GPU box: ms per attention launch for every variant

What each one removes tells what that resource costs the shipped kernel:
    half_lds   K / V fragments re-read from the LDS for every second k-step only (the LDS traffic of a 64-query wave)
    no_c       the compat stream is not fetched (c = 1): HBM / L2 traffic of the cache
    no_exp     v_exp_f32 replaced by a subtraction
    no_dma     the in-loop K / V tile refills are not issued (stale tiles)
    no_split   probabilities not split into two fp16 planes (one conversion)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gmf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab")

PATCHES = {
    "base": [],
    "half_lds": [
        ("if (pr == 0 && s < 7) { kh_n", "if (pr == 0 && s < 7 && (s & 1)) { kh_n"),
        ("if (pr == 0 && u < 21) {\n", "if (pr == 0 && u < 21 && ((u / 3) & 1)) {\n"),
    ],
    "no_c": [
        ("const f32x4 v = __builtin_nontemporal_load(ct + q * 64);", "const f32x4 v = {1.f, 1.f, 1.f, 1.f}; (void)ct;"),
    ],
    "no_exp": [
        ("          x[r] = __builtin_amdgcn_exp2f(x[r] - m_off);\n          ls += x[r];",
         "          x[r] = x[r] - m_off;\n          ls += x[r];"),
    ],
    "no_dma": [
        ("        if (u >= 3 && u < 11) issue_piece(t, u - 3);\n", ""),
    ],
    "no_split": [
        ("          split2h(x[j], x[j + 1], ph0, pl0, j);", "          ph0[j] = (_Float16)x[j]; ph0[j + 1] = (_Float16)x[j + 1];"),
        ("          split2h(x[8 + j], x[8 + j + 1], ph1, pl1, j);", "          ph1[j] = (_Float16)x[8 + j]; ph1[j + 1] = (_Float16)x[8 + j + 1];"),
    ],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    src = open(os.path.join(CSRC, "encoder_kernels.hip")).read()
    objs = [o for o in os.listdir(CSRC) if o.endswith(".o") and o != "encoder_kernels.o"]
    for name, patches in PATCHES.items():
        text = src
        for old, new in patches:
            n = text.count(old)
            if n < 1:
                raise SystemExit(f"{name}: pattern not found: {old[:50]!r}")
            text = text.replace(old, new)
        work = os.path.join(OUT, "src_" + name)
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        for f in os.listdir(CSRC):
            if f.endswith(".hpp"):
                shutil.copy(os.path.join(CSRC, f), work)
        open(os.path.join(work, "encoder_kernels.hip"), "w").write(text)
        obj = os.path.join(work, "encoder_kernels.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-fno-slp-vectorize", "-c", os.path.join(work, "encoder_kernels.hip"), "-o", obj])
        lib = os.path.join(OUT, f"libgmf_hip_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj]
                              + [os.path.join(CSRC, o) for o in objs] + ["-o", lib])
        shutil.rmtree(work)
        print("built", lib, flush=True)


def run(argv):
    for name in PATCHES:
        lib = os.path.join(OUT, f"libgmf_hip_{name}.so")
        env = dict(os.environ, GMF_LIB=lib, ROWS="2")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_times.py")] + argv, env=env,
                           capture_output=True, text=True)
        print(f"== {name}", flush=True)
        print("\n".join(l for l in r.stdout.splitlines() if "scattn" in l or "encode:" in l), flush=True)
        if r.returncode:
            print(r.stderr[-1500:])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:])
