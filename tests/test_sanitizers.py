"""CPU tier: AddressSanitizer + UndefinedBehaviorSanitizer over the host-side plan/table builders (design.hpp,
zp_tables.hpp, pz_tables.hpp, ref_plan.hpp) and the kernel bodies in lock-step emulation."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_asan_ubsan_clean():
    d = os.path.join(HERE, "emul")
    subprocess.run(["make", "-C", d, "-s", "san_check"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(d, "san_check")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizer run ok" in r.stdout
