"""The C++ host layer: container tests run anywhere, API + CLI tests need the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "edsparser_amd", "host")
BUILD = os.path.join(HOST, "build")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "edsparser_amd")


def _build_host():
    import edsparser_amd.build as b
    b.build()
    subprocess.run(["make", "-s", "-C", HOST], check=True)


def _compile(src, out):
    _build_host()
    cmd = ["g++", "-std=c++17", "-O1", "-I", INC, src, os.path.join(BUILD, "libedsparser_lib.a"),
           "-L", LIBDIR, "-ledsx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    subprocess.run(cmd, check=True)
    return out


def test_container_cpu():
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "test_container.cpp"), os.path.join(BUILD, "test_container"))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cli_argument_errors_cpu(tmp_path):
    _build_host()
    r = subprocess.run([os.path.join(BUILD, "msa2eds")], capture_output=True, text=True)
    assert r.returncode == 1 and "the option '--input' is required but missing" in r.stderr
    assert "[Performance] Runtime:" in r.stderr
    bad = tmp_path / "x.txt"
    bad.write_text(">a\nAC\n>b\nAC\n")
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Input file must be an MSA file (.msa)" in r.stderr
    eds = tmp_path / "x.eds"
    eds.write_text("{A}")
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "-i", str(eds), "-l", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Context length must be > 0" in r.stderr
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--context-length" in r.stdout


@pytest.mark.gpu
def test_cpp_api_gpu():
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "test_msa_api.cpp"), os.path.join(BUILD, "test_msa_api"))
    r = subprocess.run([exe, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_msa2eds_cli_gpu(tmp_path):
    _build_host()
    msa = tmp_path / "small.msa"
    msa.write_bytes(open(os.path.join(ROOT, "tests", "golden", "ref_data", "msa", "small.msa"), "rb").read())
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(msa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "MSA → EDS transformation" in r.stdout and "Transformation complete!" in r.stdout
    assert (tmp_path / "small.eds").read_text() == "{AGTC}{,CC}{T}{C,A}{TATAAAT}{AA,GG}{ATA}{,GGGG}"
    assert (tmp_path / "small.seds").read_text() == "{0}{1,3}{2}{0}{1}{2,3}{0}{1,2}{3}{0}{1,3}{2}"
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(msa), "-l", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "small_l4.leds").read_text() == "{AGTC}{TC,CCTA,TA}{TATAAAT}{AAATA,AAATAGGGG,GGATA}"
    assert (tmp_path / "small_l4.seds").read_text() == "{0}{1}{2}{3}{0}{1}{2}{3}"


@pytest.mark.gpu
def test_vcf2eds_and_eds2leds_cli_gpu(tmp_path):
    _build_host()
    g = os.path.join(ROOT, "tests", "golden", "ref_data")
    for f in ("vcf/small.vcf", "vcf/small.fa", "eds/test_iterative.eds"):
        (tmp_path / os.path.basename(f)).write_bytes(open(os.path.join(g, f), "rb").read())
    r = subprocess.run([os.path.join(BUILD, "vcf2eds"), "-i", str(tmp_path / "small.vcf"), "-r", str(tmp_path / "small.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "small.eds").read_bytes() == open(os.path.join(g, "vcf/small.eds"), "rb").read()
    assert (tmp_path / "small.seds").read_bytes() == open(os.path.join(g, "vcf/small.seds"), "rb").read()
    assert "Variant groups created:     10" in r.stdout and "Success rate:               100.0%" in r.stdout
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "-i", str(tmp_path / "test_iterative.eds"), "-l", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "test_iterative_l4.leds").read_bytes() == open(os.path.join(g, "eds/test_iterative_l4.eds"), "rb").read()
    assert "Output mode: compact" in r.stdout and "Threads: 1 (sequential)" in r.stdout
