"""Host-side sanitizer pass ("race detection / sanitizers"): every source of the library compiled
HOST-ONLY (`hipcc --cuda-host-only`: the kernels become launch stubs) with AddressSanitizer + UBSan and, separately, ThreadSanitizer,
linked against a stand-in HIP runtime (tests/host_sanitize/hip_stub.cc: device memory = host memory, kernels do nothing) and driven by
tests/host_sanitize/driver.cc -- six caller threads with rotating pipeline slots submitting steps to the library's two launch
threads, synchronous steps, phase-1 redos, geometry epochs, the list route, a custom communicator with pieces, option changes and
workspace releases while steps are in flight.  No GPU, no GPU sanitizer: the device code is not compiled at all."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "torch-assimilate_amd", "csrc")
HERE = os.path.join(ROOT, "tests", "host_sanitize")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build(tmp, san_flags):
    sys.path.insert(0, os.path.join(ROOT, "torch-assimilate_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mia_build", os.path.join(ROOT, "torch-assimilate_amd", "_build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    common = ["-O1", "-g", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + san_flags

    def cc(src):
        obj = os.path.join(tmp, os.path.basename(src) + ".o")
        cmd = [HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-x", "hip"] + common + bld.SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
        return obj

    srcs = [os.path.join(CSRC, s) for s in bld.SOURCES]
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, srcs))
    # the fat-binary symbols the host objects expect from the (skipped) device pass
    nm = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
    fat = sorted({l.split()[-1] for l in nm.splitlines() if "__hip_fatbin_" in l})
    with open(os.path.join(tmp, "fatbin_stub.cc"), "w") as fh:
        fh.write("".join('extern "C" { extern const char %s[16]; const char %s[16] = {0}; }\n' % (s, s) for s in fat))
    exe = os.path.join(tmp, "driver")
    rocm_inc = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "include")
    cmd = [HIPCC, "--cuda-host-only", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I" + rocm_inc] + common + \
          [os.path.join(HERE, "hip_stub.cc"), os.path.join(HERE, "driver.cc"), os.path.join(tmp, "fatbin_stub.cc"), "-x", "none"] + objs + \
          ["-ldl", "-pthread", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def _sanitizer_runtime(name):
    import glob
    return bool(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.a" % name))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name,flags,env", [
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
                                                                                  "LSAN_OPTIONS": "suppressions=" + os.path.join(HERE, "lsan.supp") + ":print_suppressions=0"}),
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"}),
])
def test_step_driver_host_logic_under_sanitizers(tmp_path, name, flags, env):
    if not _sanitizer_runtime(name):
        pytest.skip("clang's %s runtime is not installed" % name)
    exe = _build(str(tmp_path), flags)
    res = subprocess.run([exe, "6", "60"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-6000:]
    assert "failed checks" in out and ", 0 failed checks" in out, out[-3000:]
    for marker in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:", "ERROR: LeakSanitizer"):
        assert marker not in out, out[-6000:]
