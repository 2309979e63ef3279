"""Compile every HIP source with -save-temps into a scratch directory and list each kernel's
VGPRs, LDS and scratch. A kernel with scratch (spills / dynamically indexed arrays) pays tens of
microseconds of dispatch-time scratch setup on this stack: the latency-path kernels must have none.
Usage: python scripts/kernel_resources.py [file.hip ...]   (exit code 1 if any kernel uses scratch)"""
import glob, os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "voitta_rag_amd/csrc/*.hip")))
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for f in src:
        base = os.path.splitext(os.path.basename(f))[0]
        subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "--offload-arch=gfx950",
                        "-I" + os.path.join(root, "include"), "-c", os.path.abspath(f), "-o", base + ".o", "-save-temps"],
                       cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        for blk in asm.split("  - .agpr_count")[1:]:
            get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            scratch = get("private_segment_fixed_size")
            bad += scratch > 0
            print(f"{base:8s} {name[:58]:58s} vgpr {get('vgpr_count'):4d} lds {get('group_segment_fixed_size'):7d} "
                  f"scratch {scratch:5d}{'   <-- SCRATCH' if scratch else ''}")
sys.exit(1 if bad else 0)
