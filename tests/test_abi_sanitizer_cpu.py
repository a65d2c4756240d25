"""Host half of the C ABI under AddressSanitizer + UBSan (no GPU needed): `make -C eav_amd/csrc -f asan.mk` builds
libeav_hip_asan.so (host code instrumented, device code plain - GPU ASan is not available on this pool); a child process
preloads the ASan runtime, loads the library and drives every entry point on the paths that return before anything would
be launched - the plan / size helpers over a grid of shapes (integer arithmetic, divisions, clamps) and every status
function with null / degenerate arguments (argument validation, error-string formatting).  Any report fails the test."""
import glob
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, itertools, sys
sys.path.insert(0, ROOT)
from eav_amd import _lib
lib = C.CDLL(LIBPATH)
n_plain = n_status = 0
for name, (args, res) in _lib.PLAIN.items():
    fn = getattr(lib, name)
    fn.argtypes, fn.restype = args, res
    if not args:
        fn()
        n_plain += 1
        continue
    ints = [a for a in args]
    grids = [(1, 2, 7, 8, 30, 64, 197, 768, 1214, 3072, 9712, 25216, 10000)] * len(args)
    for k, vals in enumerate(itertools.islice(itertools.product(*grids), 0, 4000, 37)):
        fn(*[a(v) for a, v in zip(args, vals)])
        n_plain += 1
lib.eav_last_error.restype = C.c_char_p
for name, args in _lib.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.argtypes, fn.restype = args, C.c_int
    if name in ("eav_gemm_sp_set_tile", "eav_sp_set_convert_blocks", "eav_attn_sp_set_nw4_above"):
        fn(*[a(0) for a in args])                  # tuning hooks: host state only
        n_status += 1
        continue
    if not any(a is C.c_void_p for a in args):
        continue                                   # nothing to invalidate: would launch
    for fill in (0, -1, 1):                        # all pointers NULL, sizes zero / negative / one
        vals = [None if a is C.c_void_p else a(fill) for a in args]
        rc = fn(*vals)
        assert rc != 0, (name, fill, "accepted NULL pointers")
        assert lib.eav_last_error(), name
        n_status += 1
print("asan-ok", n_plain, n_status)
"""


def test_host_half_of_the_abi_is_clean_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "eav_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-f", "asan.mk", "-j8", "asan"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    libpath = os.path.join(ROOT, "eav_amd", "libeav_hip_asan.so")
    rts = sorted(glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rts, "ASan runtime of the ROCm clang not found"
    script = tmp_path / "child.py"
    script.write_text(f"ROOT = {ROOT!r}\nLIBPATH = {libpath!r}\n" + textwrap.dedent(CHILD))
    env = dict(os.environ, LD_PRELOAD=rts[-1],
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "asan-ok" in r.stdout, out[-6000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
