"""CPU: the MATLAB boundary (matlab/*.m + matlab/ekfslam_mex.c), as far as it can be checked without MATLAB.

There is no MATLAB / Octave / MathWorks mex.h in the image, so none of this runs under MATLAB.  What runs:
  1. the gateway type-checks against include/ekfslam.h (gcc -fsyntax-only, declarations-only MEX API subset);
  2. the gateway is COMPILED (ASan + UBSan) against a small mock of that MEX API subset and a recording stand-in for libekfslam
     (tests/support/mex_mock/) and every command is driven with the argument shapes the .m classes pass -- including the `[]`
     handle of EKF_SLAM.f, empty / wrong-typed / null handles and a failing ABI call; the transcript is checked here;
  3. every ekfslam_mex('cmd', ...) call site in matlab/*.m uses a (command, argument count) the driver exercised;
  4. every member of the replaced classes that the reference's own callers use (tests/golden/reference_call_sites.json,
     extracted from SLAM.m / test_slam_class.m / EKF_SLAM_UC.m by tests/golden/make_call_sites.py) exists in matlab/*.m with a
     compatible argument count, and the reference's public properties are assignable there too.
The mock pins the gateway's own logic; it says nothing about MATLAB itself."""
import glob
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "support", "mex_mock")
INCLUDES = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "support", "mex_api_subset"), "-I", MOCK]


def test_mex_gateway_type_checks_against_the_abi():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=implicit-function-declaration",
                        "-Werror=incompatible-pointer-types", "-Werror=int-conversion"] + INCLUDES +
                       [os.path.join(ROOT, "matlab", "ekfslam_mex.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def _matlab_call_sites():
    """(file, command, nrhs) of every ekfslam_mex('command', ...) in matlab/*.m (continuation lines joined)."""
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "matlab", "*.m"))):
        text = re.sub(r"\.\.\.[^\n]*\n", " ", open(f).read())
        for m in re.finditer(r"ekfslam_mex\(\s*'([A-Za-z_]+)'", text):
            depth, nargs, i = 0, 1, m.start() + len("ekfslam_mex")
            while True:
                ch = text[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                elif ch == "," and depth == 1:
                    nargs += 1
                i += 1
            out.append((os.path.basename(f), m.group(1), nargs))
    return out


def test_matlab_classes_only_use_commands_the_gateway_implements():
    used = {c for _, c, _ in _matlab_call_sites()}
    impl = set(re.findall(r'strcmp\(cmd, "([A-Za-z_]+)"\)', open(os.path.join(ROOT, "matlab", "ekfslam_mex.c")).read()))
    assert used and used <= impl, sorted(used - impl)


@pytest.fixture(scope="module")
def transcript(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mexmock") / "mexdrv")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-g", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined"] + INCLUDES +
                       [os.path.join(ROOT, "matlab", "ekfslam_mex.c")] +
                       [os.path.join(MOCK, f) for f in ("mex_mock.c", "abi_stub.c", "driver.c")] + ["-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, "the gateway crashed under the mock:\n" + r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout.splitlines()


def _after(lines, mex_prefix, nth=0):
    """ABI lines logged between the previous MEX line and the nth line starting with mex_prefix, plus that line."""
    hits = [i for i, ln in enumerate(lines) if ln.startswith(mex_prefix)]
    i = hits[nth]
    j = i - 1
    while j >= 0 and lines[j].startswith("ABI "):
        j -= 1
    return lines[j + 1:i], lines[i]


def test_gateway_marshalling_under_the_mock(transcript):
    t = transcript
    # f: no handle needed, [] accepted, both outputs shaped like the reference's [x_new, F]
    abi, mex = _after(t, "MEX f nrhs=4")
    assert abi == ["ABI ekf_motion_model n=5 x0=1 u=0.1,3 F=yes"] and "out0=1x5[101," in mex and "out1=5x5[" in mex
    # create: optional tile / batch reach the config; a failing create destroys the half-made handle and raises
    assert "ABI ekf_create mode=1 cap=8 tile=64 batch=4" in t
    i = t.index("ABI ekf_create mode=1 cap=666 tile=0 batch=1")
    assert t[i + 1] == "ABI ekf_destroy" and t[i + 2].startswith("MEX create nrhs=3 -> ERROR ekfslam:status")
    # 1-based MATLAB indices become 0-based at the ABI and back
    assert "ABI ekf_correct z=5,50 R=0.05,0,0,250 idx0=0" in t
    assert any(ln.startswith("MEX associate") and "out0=1x1L[0] out1=1x1[7]" in ln for ln in t)
    assert "ABI ekf_get_P_block r0=3 c0=3 nr=2 nc=2" in t
    # measure: m from the row count of observed_LL, L from the index vector, column-major pass-through
    assert "ABI ekf_measure m=2 obs_r0=5,40,1 obs_last=2 u=0.1,3 L=3 idx0=1 loc0=10,20" in t
    # set_state orders x, s, P (s and P need the landmark count x fixes)
    abi, _ = _after(t, "MEX set_state")
    assert [a.split()[1] for a in abi] == ["ekf_set_x", "ekf_set_s", "ekf_set_P"]
    assert any(ln.startswith("MEX get_Q") and "out0=5x5[10,11,12,0,0,13]" in ln for ln in t)       # zeros(size(P)) + 3x3 block
    assert any(ln.startswith("MEX get_P_diag_blocks") and "out0=4x2[" in ln for ln in t)
    # errors: a failing status carries ekf_last_error; handles are validated, never dereferenced blindly
    assert "MEX predict nrhs=3 -> ERROR ekfslam:status | call not valid in the current state: injected failure" in t
    bad = [ln for ln in t if "ERROR ekfslam:handle" in ln]
    assert len(bad) == 7                                  # [] handle, double handle, null uint64, missing handle; exchange_local: doubles, a null inside, []
    # several GPUs from one host thread (matlab/ShardedEKF.m): device / rank / world reach the config; begin on every shard,
    # ONE exchange over the handle vector in shard order, finish on every shard; 1-based index converted once
    assert "ABI ekf_create mode=1 cap=32 tile=0 batch=1 device=1 rank=1 world=2" in t
    i = t.index("ABI ekf_exchange_local world=2 ranks=0,1")
    assert [ln.split()[1] for ln in t[i - 4:i + 5] if ln.startswith("ABI ")] == \
        ["ekf_correct_begin", "ekf_correct_begin", "ekf_exchange_local", "ekf_correct_finish", "ekf_correct_finish"]
    assert "ABI ekf_correct_begin rank=1 z=5,50 R=0.05,0,0,250 idx0=2" in t
    assert "ABI ekf_hint_next rank=1 idx0=6" in t                                  # ekf_hint_next: 1-based -> 0-based once
    assert "ABI ekf_associate_begin rank=0 z=5,50,7 R=0.05,0,0,250 costs=0" in t
    assert any(ln.startswith("MEX associate_finish") and "out0=1x1L[0] out1=1x1[5]" in ln for ln in t)
    i = t.index("ABI ekf_create storage=1 pass_arith=1")                            # create's 9th / 10th argument: cfg.storage, cfg.pass_arith
    assert t[i + 1] == "ABI ekf_create mode=0 cap=64 tile=256 batch=32"
    assert any(ln.startswith("MEX create nrhs=6 -> ERROR ekfslam:usage") for ln in t)
    assert any("ERROR ekfslam:usage | 'predict' needs 3 arguments" in ln for ln in t)
    assert any("unknown command 'no_such_command'" in ln for ln in t)
    assert t[-2:] == ["LOCKS 0", "MISUSE 0"]              # create/destroy balance mexLock; no mxGetScalar on an empty array


def test_every_m_call_site_shape_is_exercised_by_the_driver(transcript):
    ok = {(m.group(1), int(m.group(2))) for m in (re.match(r"MEX (\w+) nrhs=(\d+) -> ok", ln) for ln in transcript) if m}
    missing = [(f, c, n) for f, c, n in _matlab_call_sites() if (c, n) not in ok]
    assert not missing, "call shapes in matlab/*.m the mock driver never ran: %s" % missing


def test_no_command_reads_the_handle_before_it_is_known_to_be_one():
    src = open(os.path.join(ROOT, "matlab", "ekfslam_mex.c")).read()
    body = src[src.index("void mexFunction("):]
    at = body.index("handle_of(nrhs, prhs)")
    before = body[:at]
    # the only commands dispatched before the handle is validated are the two that take none and the one that takes a VECTOR
    # of handles, and each of them returns
    assert re.findall(r'strcmp\(cmd, "([A-Za-z_]+)"\)', before) == ["create", "f", "exchange_local"]
    assert before.count("return;") >= 3
    # create / f never turn prhs[1] into a pointer (create reads it as the numeric mode argument: mxGetScalar only) ...
    xl = before.index('strcmp(cmd, "exchange_local")')
    assert not re.search(r"mxGet(Data|Pr)\(prhs\[1\]\)", before[:xl])
    # ... and exchange_local checks class and data pointer before it reads the vector, and every element before it is used
    blk = before[xl:]
    assert blk.index("mxGetClassID(prhs[1]) != mxUINT64_CLASS") < blk.index("(const uint64_t *)mxGetData(prhs[1])")
    assert blk.index("if (!hs[r])") < blk.index("ekf_exchange_local(hs")
    assert "mxGetData(prhs[1])" not in body[at:].replace("handle_of(nrhs, prhs)", "")    # after that only handle_of() touches it


def _m_class(name):
    """(methods: {name: n_declared_args incl. the object}, properties: set, settable: set) of matlab/<name>.m, with EKF_SLAM's
    members inherited by EKF_SLAM_UC."""
    text = open(os.path.join(ROOT, "matlab", name + ".m")).read()
    methods, props, setters = {}, set(), set()
    for m in re.finditer(r"^\s*function\s+(?:\[?[\w,\s~]*\]?\s*=\s*)?([\w.]+)\s*\(([^)]*)\)", text, re.M):
        fname, args = m.group(1), [a for a in m.group(2).split(",") if a.strip()]
        if fname.startswith("set."):
            setters.add(fname[4:])
        elif not fname.startswith("get."):
            methods[fname] = len(args)
    for blk in re.finditer(r"properties[^\n]*\n(.*?)\n\s*end", text, re.S):
        head = text[blk.start():text.index("\n", blk.start())]
        for pm in re.finditer(r"([A-Za-z_]\w*)\s*(?:=[^;]*)?;", blk.group(1)):
            props.add(pm.group(1))
            if "Dependent" not in head and "protected" not in head:
                setters.add(pm.group(1))
    if re.search(r"classdef\s+%s\s*<\s*EKF_SLAM\b" % name, text):
        bm, bp, bs = _m_class("EKF_SLAM")
        methods = dict(bm, **methods); props |= bp; setters |= bs
    return methods, props, setters


def test_matlab_classes_offer_every_member_the_references_callers_use():
    fixture = os.path.join(ROOT, "tests", "golden", "reference_call_sites.json")
    sites = json.load(open(fixture))["sites"]
    if os.path.isdir("/root/reference"):            # build container: the fixture must be what the generator extracts today
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_call_sites", os.path.join(ROOT, "tests", "golden", "make_call_sites.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.extract("/root/reference") == sites
    assert {s["member"] for s in sites} >= {"predict", "measure", "plot", "x", "estimateCorrespondence", "<constructor>"}
    for s in sites:
        for cls in s["receiver_class"].split("|"):
            methods, props, _ = _m_class(cls)
            where = "%s:%d %s.%s" % (s["file"], s["line"], cls, s["member"])
            if s["member"] == "<constructor>":
                text = open(os.path.join(ROOT, "matlab", cls + ".m")).read()
                # a constructor called with fewer arguments than declared must default them (nargin / varargin)
                declared = methods.get(cls, 0)
                assert s["paren_args"] <= declared or "varargin" in text, where
                assert s["paren_args"] == declared or "nargin" in text or "varargin" in text, where
            elif s["member"] in methods:
                assert s["paren_args"] + 1 == methods[s["member"]], where + ": argument count differs from the reference's call"
            else:
                assert s["member"] in props, where + " is neither a method nor a property of matlab/%s.m" % cls


def test_reference_public_properties_are_assignable():
    """EKF_SLAM.m:5-22 / EKF_SLAM_UC.m:5-22 declare x P Q s C Rc ... as plain (assignable) properties."""
    for cls, want in (("EKF_SLAM", {"x", "P", "Q", "s", "C", "Rc", "s_cost", "s_thresh", "landmark_list", "observed"}),
                      ("EKF_SLAM_UC", {"x", "P", "Q", "s", "C", "Rc", "correspondence", "landmark_list", "observed"})):
        _, props, setters = _m_class(cls)
        assert want <= props, sorted(want - props)
        assert want <= setters, "not assignable in matlab/%s.m: %s" % (cls, sorted(want - setters))


def test_correspondence_forwards_its_own_cost_and_threshold():
    text = open(os.path.join(ROOT, "matlab", "Correspondence.m")).read()
    m = re.search(r"ekfslam_mex\('set_params',\s*tmp,[^\n]*\n", text)
    assert m and "h.s_cost" in m.group(0) and "h.s_thresh" in m.group(0)
    assert text.index("'set_params'") < text.index("'associate'")


def test_landmark_selector_offers_the_synthetic_source():
    """matlab/Landmark.m keeps the reference's surface (Landmark.m:12-33: Landmark(method), .landmarkObj, .method, getLandmark(laserdata, x))
    and adds 'SYNTHETIC' -> SyntheticLandmarks, whose .landmark is the struct array of RANSAC.m:238-241 that measure() indexes
    (EKF_SLAM.m:111,119).  Static checks only: MATLAB is not in the image; the Python twin (ekf_slam_amd/world.py::SyntheticLandmark)
    is what runs, on the GPU, against the oracle's."""
    methods, props, _ = _m_class("Landmark")
    assert methods.get("Landmark") == 1 and methods.get("getLandmark") == 3 and {"landmarkObj", "method"} <= props
    text = open(os.path.join(ROOT, "matlab", "Landmark.m")).read()
    m = re.search(r"case\s+'SYNTHETIC'\s*\n\s*h\.landmarkObj\s*=\s*(\w+)\(\)", text)
    assert m and m.group(1) == "SyntheticLandmarks"
    assert re.search(r"case\s+'RANSAC'\s*\n\s*h\.landmarkObj\s*=\s*RANSAC\(\)", text)          # the reference's own source stays selectable
    smethods, sprops, _ = _m_class("SyntheticLandmarks")
    assert smethods.get("getLandmark") == 3 and smethods.get("plot") == 3 and "landmark" in sprops
    stext = open(os.path.join(ROOT, "matlab", "SyntheticLandmarks.m")).read()
    for field in ("loc", "observe", "index", "fresh"):                                           # RANSAC.m:238-241
        assert re.search(r"h\.landmark\(at\)\.%s\s*=" % field, stext), field
    assert "sortrows(observed_LL, 3)" in stext                                                   # rows in index order, as the Python twin returns them
    # what measure() reads from the source exists on it: .landmarkObj.landmark with .index / .loc
    e = open(os.path.join(ROOT, "matlab", "EKF_SLAM.m")).read()
    assert "landmark_list.landmarkObj.landmark" in e and "landmark_list.getLandmark(laserData, h.x)" in e


def test_synthetic_slam_facade_keeps_the_references_surface():
    """matlab/SyntheticSLAM.m: SLAM.m's facade (SLAM.m:17-68,105-116) with the ROS subscribers replaced by a feed -- the members the
    reference's drivers touch on a SLAM object exist with the reference's argument counts, and every member it calls on h.slam / h.LM
    exists on matlab/EKF_SLAM.m / Landmark.m."""
    methods, props, _ = _m_class("SyntheticSLAM")
    assert {"LM", "slam", "algorithmName", "u"} <= props
    assert methods.get("predict") == 2 and methods.get("measure") == 3 and methods.get("plot") == 1 and methods.get("runSlam") == 1
    text = open(os.path.join(ROOT, "matlab", "SyntheticSLAM.m")).read()
    assert "EKF_SLAM(varargin{:})" in text and "EKF_SLAM_UC(varargin{:})" in text and "Landmark(landmark_method)" in text
    em, _, _ = _m_class("EKF_SLAM")
    for member, nargs in re.findall(r"h\.slam\.(\w+)\(([^)]*)\)", text):
        assert member in em and em[member] == 1 + len([a for a in nargs.split(",") if a.strip()]), member
