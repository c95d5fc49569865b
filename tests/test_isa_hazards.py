"""CPU: the HAZARD RULE of csrc/query.hip checked on the compiled ISA (tools/check_isa.py): no inline-asm statement inside the hazard window of a matrix
instruction's destination registers.  hipcc cross-compiles gfx950 without a GPU; the reproducer proves the checker sees the failure it guards against."""
import os
import subprocess
import sys

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


def _isa(src, out, flags=()):
    subprocess.check_call([HIPCC, "-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *flags, "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)


def test_checker_flags_the_reproducer(tmp_path):
    out = str(tmp_path / "bad.s")
    _isa(os.path.join(ROOT, "tools", "bench_scripts", "asm_hazard_repro.hip"), out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), out], capture_output=True, text=True)
    assert r.returncode == 1 and "VIOLATION" in r.stdout and "_Z3bad" in r.stdout and "_Z4good" not in r.stdout.split("VIOLATION", 1)[1], r.stdout


def test_query_kernels_keep_the_hazard_rule(tmp_path):
    out = str(tmp_path / "query.s")
    _isa(os.path.join(ROOT, "vistracker_amd", "csrc", "query.hip"), out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), out], capture_output=True, text=True)
    assert r.returncode == 0 and "0 asm statements inside the hazard window" in r.stdout, r.stdout
