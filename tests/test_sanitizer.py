"""The product's host-side file readers under AddressSanitizer + UBSan, on the CPU (no GPU, no HIP code involved).

hu_host.cpp (`.hmm`, `.ptu`) and hu_seedindex.cpp (`.csfm`, the reference's libcds layouts) parse files a user hands to the library;
they are compiled here with `g++ -fsanitize=address,undefined` next to tests/san/san_driver.cpp, which feeds them hundreds of damaged
copies of valid files — truncations, byte flips, length fields overwritten with huge or negative values.  A reader may accept or
refuse a damaged file; it may not read or write outside its buffers, overflow an index, or size an allocation by a number the file
cannot back (the sanitizer aborts on each of those and the test fails).  GPU sanitizers are not available on the pool; this is the
CPU half the task statement asks for."""
import gzip, os, shutil, subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hmmufotu_amd", "csrc")
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("san")
    exe = str(d / "san_driver")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + CSRC, "-o", exe,
           os.path.join(ROOT, "tests", "san", "san_driver.cpp"), os.path.join(CSRC, "hu_host.cpp"), os.path.join(CSRC, "hu_seedindex.cpp"), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe, d


def _run(exe, args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:] + "\n" + "\n".join(l for l in r.stderr.splitlines() if not l.startswith(("  0x", "=>")))[-4000:])
    return r.stdout


def test_hmm_and_ptu_readers_on_damaged_files(driver):
    from hmmufotu_amd import synth
    exe, d = driver
    db = synth.make_db(24, 400, "GTR", dg_k=4)
    hp, pp = str(d / "t.hmm"), str(d / "t.ptu")
    synth.write_hmm(db.hmm, hp); synth.write_ptu(db, pp)
    out = _run(exe, ["hmm", hp, d / "scratch.hmm", 1500])
    assert "1500 trials" in out
    out = _run(exe, ["ptu", pp, d / "scratch.ptu", 600])
    assert "600 trials" in out and " 0 refused" not in out


def test_csfm_reader_on_damaged_files(driver):
    exe, d = driver
    p = d / "awkward.csfm"
    p.write_bytes(gzip.open(os.path.join(G, "csfm_awkward.csfm.gz"), "rb").read())
    out = _run(exe, ["csfm", p, d / "scratch.csfm", 1000, 240, 12])
    assert "1000 trials" in out and " 0 refused" not in out
    p = d / "70.csfm"
    p.write_bytes(gzip.open(os.path.join(G, "70_otus.csfm.gz"), "rb").read())
    out = _run(exe, ["csfm", p, d / "scratch.csfm", 40, 1486, 20])
    assert "40 trials" in out
