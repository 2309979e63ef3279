"""TEST INFRASTRUCTURE — compiles oracle_core.c with gcc into oracle/_build/ (git-ignored)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle_core.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "oracle_core.c")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", LIB, src, "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
