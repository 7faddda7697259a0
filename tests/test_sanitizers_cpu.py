"""CPU sanitizer runs ("host-side ASan/UBSan build of the CPU restatement"; GPU ASan / xnack+ do not exist on the pool).

* `make -C oracle asan`  -> build_asan/libfusion_oracle_asan.so (gcc -fsanitize=address,undefined): the oracle's golden-vector tests are
  re-run through it in a child process (LD_PRELOAD of gcc's libasan into an uninstrumented python).
* `make -C fusion_amd/csrc hostasan` -> build_asan/libfusion_hip_hostasan.so: the HOST side of every C-ABI entry point (argument
  validation, workspace planning, launch set-up) compiled with clang's ASan + UBSan, device code untouched; the no-GPU ABI tests are
  re-run against it (FUSION_AMD_LIB) with clang's runtime preloaded.
Any sanitizer report aborts the child (halt_on_error), which fails the test."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ASAN_DIR = os.path.join(ROOT, "build_asan")


def _run(cmd, env_extra, timeout):
    env = dict(os.environ)
    env.update(ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **env_extra)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    return p.returncode, p.stdout


def test_oracle_golden_vectors_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "-s"])
    so = os.path.join(ASAN_DIR, "libfusion_oracle_asan.so")
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc has no libasan.so in this image")
    rc, out = _run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "tests/test_oracle_golden_r2.py", "-x", "-q", "-p", "no:cacheprovider"],
                   dict(LD_PRELOAD=rt, FUSION_ORACLE_SO=so), timeout=900)
    assert rc == 0 and " passed" in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]


def test_c_abi_host_side_under_asan_ubsan():
    """Every exported symbol binds, and the argument-validation / workspace-planning paths run clean, in the host-sanitized build."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "fusion_amd", "csrc"), "hostasan", "-j4", "-s"])
    so = os.path.join(ASAN_DIR, "libfusion_hip_hostasan.so")
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.exists(rt):
        pytest.skip("clang has no shared ASan runtime in this image")
    tests = ["tests/test_host_cpu.py::test_library_loads_and_exports_header_symbols", "tests/test_host_cpu.py::test_abi_argument_validation_without_gpu",
             "tests/test_host_cpu.py::test_integration_md_binding_matches_the_abi", "tests/test_host_cpu.py::test_workspace_planning_is_consistent",
             "tests/test_host_cpu_r4.py::test_tables_abi_planning_and_validation_without_gpu"]
    rc, out = _run([sys.executable, "-m", "pytest", *tests, "-x", "-q", "-p", "no:cacheprovider"], dict(LD_PRELOAD=rt, FUSION_AMD_LIB=so), timeout=900)
    assert rc == 0 and "5 passed" in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]


def test_no_mfma_result_is_read_before_it_is_written(tmp_path):
    """tools/check_mfma_hazards.py over the gfx950 assembly of every kernel file that issues MFMAs: hipcc pads 'MFMA write -> read' inside a
    basic block; across a branch it once left the first reader unpadded (attn_varlen_kernel's tile maximum: results correct but different
    from run to run).  The kernels are written so that no reader sits behind a branch unpadded; this keeps it that way (cross-compiles, no GPU)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "fusion_amd", "csrc")
    files = [f for f in sorted(os.listdir(src)) if f.endswith(".hip") and "mfma" in open(os.path.join(src, f)).read()]
    assert {"score.hip", "maxsim.hip", "encoder.hip"} <= set(files)
    outs = []
    for f in files:
        out = str(tmp_path / (f[:-4] + ".s"))
        subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                        os.path.join(src, f), "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"), *outs], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
