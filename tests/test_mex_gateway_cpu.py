"""CPU: matlab/ekfslam_mex.c type-checks against include/ekfslam.h.  There is no MATLAB here, so the gateway cannot be built
or run; this compiles it with -fsyntax-only against a declarations-only subset of the documented MEX C API
(tests/support/mex_api_subset/mex.h), which catches a gateway call whose arguments no longer match the C ABI."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mex_gateway_type_checks_against_the_abi():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=implicit-function-declaration",
                        "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "support", "mex_api_subset"),
                        os.path.join(ROOT, "matlab", "ekfslam_mex.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_matlab_classes_only_use_commands_the_gateway_implements():
    import glob
    import re
    used = set()
    for f in glob.glob(os.path.join(ROOT, "matlab", "*.m")):
        used |= set(re.findall(r"ekfslam_mex\('([A-Za-z_]+)'", open(f).read()))
    impl = set(re.findall(r'strcmp\(cmd, "([A-Za-z_]+)"\)', open(os.path.join(ROOT, "matlab", "ekfslam_mex.c")).read()))
    assert used and used <= impl, sorted(used - impl)
