"""Build recipe for the oracle's C restatement (TEST INFRASTRUCTURE ONLY).

``gcc -O2 -ffp-contract=off`` on ``oracle/csrc/march.c`` -> ``oracle/_build/libren_oracle.so``.
There is no ``oracle/_ref``: the reference is pure Python with CUDA-only third-party
kernels, so nothing of it can be compiled here (DESIGN.md, section "Oracle").
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "march.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libren_oracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.check_call([
            "gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
            SRC, "-o", OUT, "-lm",
        ])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
