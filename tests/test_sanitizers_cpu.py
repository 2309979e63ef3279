"""The host-only C++ (WordPiece, BM25 text pipeline with the Porter2 stemmer, chunker, fusion: the code that takes
untrusted document text) under AddressSanitizer + UndefinedBehaviorSanitizer: `make asan` builds
libvoitta_host_asan.so without any HIP code, and the existing CPU fuzz / known-answer tests of those modules run
against it in a child process (SURVEY.md §5 asks for sanitizers on the CPU side; GPU ASan is not available here)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_code_is_clean_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, "voitta_rag_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asan", "-j4"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "voitta_rag_amd", "libvoitta_host_asan.so")
    asan_rt = subprocess.run(["g++", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.exists(lib) and os.path.isabs(asan_rt) and os.path.exists(asan_rt), (lib, asan_rt)
    env = dict(os.environ, LD_PRELOAD=asan_rt, VOITTA_ENGINE_LIB=lib, VOITTA_ENGINE_HOST_ONLY="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    tests = ["tests/test_wordpiece_cpu.py", "tests/test_bm25_text_cpu.py", "tests/test_chunking_cpu.py",
             "tests/test_fusion_cpu.py", "tests/test_oracle_bm25_cpu.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *tests], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert " passed" in p.stdout
