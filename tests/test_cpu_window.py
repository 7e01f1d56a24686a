"""Host-side check of the operand-window arithmetic behind the fault fixed in round 3: a 16-byte unit of a
k-contiguous 16-bit operand can straddle the end of the contraction, and behind the last row of the last item there is no
memory.  The kernels (gemm_bf16x3.hip) and this test use the same functions (csn_amd/csrc/csn_window.h); the C++ model
(tests/host/window_test.cpp) walks every unit of 700+ geometries — contraction lengths with K % 8 in {0, 4}, row tiles that end
the buffer, contractions that fill the row — through the hardware's range check and asserts that nothing outside the allocation
is touched and that exactly the operand's values (k < K) or zeros (k >= K) reach the product."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, extra=()):
    exe = str(tmp_path / "window_test")
    subprocess.run(["g++", "-O1", "-std=c++17", *extra, os.path.join(ROOT, "tests", "host", "window_test.cpp"), "-o", exe], check=True)
    return subprocess.run([exe], capture_output=True, text=True)


def test_window_ends_with_the_last_rows_contraction(tmp_path):
    res = _run(tmp_path)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "0 violations" in res.stdout


def test_the_model_catches_the_round_3_fault(tmp_path):
    """The same walk with the window the faulting code used (whole rows: rows * ld) must report touches outside the allocation —
    otherwise the test above proves nothing."""
    res = _run(tmp_path, ["-DCSN_TEST_FULL_ROW_WINDOW"])
    assert res.returncode != 0 and "touch outside allocation" in res.stdout


def test_kernels_use_the_tested_functions():
    src = open(os.path.join(ROOT, "csn_amd", "csrc", "gemm_bf16x3.hip")).read()
    assert src.count("csn_kwin_bytes(") >= 4 and src.count("csn_unit_upper_half_beyond(") >= 2 and "#include \"csn_window.h\"" in src
