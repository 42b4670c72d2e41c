"""The launch-merging knobs of the CIN path (FIL_CIN_{HEADFOLD,FWDQ,DZ2,QMERGE,QTAIL}=0 select the older, un-merged launch sequences)
are read ONCE when the library is first used, so each setting needs a process of its own: every knob gets a fresh child that runs
the mode-64 column of test_cin_fused_tail (the merged quadratic tail forced at every tail shape, 19 cases against the fp64 closed
forms) with that knob off.  The fallbacks therefore stay part of what `pytest -m gpu` proves, not of a script somebody has to
remember to run (tools/gpu_knobs.sh runs the full 200-case subset).  The child is started before it touches the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["FIL_CIN_DWFOLD4", "FIL_CIN_PACKFOLD", "FIL_CIN_HEADFOLD", "FIL_CIN_FWDQ", "FIL_CIN_DZ2", "FIL_CIN_QMERGE", "FIL_CIN_QTAIL"])
def test_cin_parity_subset_with_one_knob_off(knob):
    env = dict(os.environ)
    env[knob] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py::test_cin_fused_tail", "-x", "-q", "-m", "gpu", "-k", "1-64-",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-15:])
    assert r.returncode == 0, "%s=0:\n%s" % (knob, tail)
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["FIL_ATTN_KREG", "FIL_ATTN_DX_LDS"])
def test_attn_parity_subset_with_one_knob_off(knob):
    """The attention kernels' forms behind knobs: FIL_ATTN_KREG=0 = the forward with its k image in LDS at the shapes whose k fragments
    fit registers (up to 13 key tiles, f16 mode), FIL_ATTN_DX_LDS=0 = the backward's dx part through global memory (second visit).
    A fresh child runs the two-waves-per-head shapes (F = 170..300, both precisions) against the closed forms."""
    env = dict(os.environ)
    env[knob] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py::test_attn_two_waves_per_head", "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-15:])
    assert r.returncode == 0, "%s=0:\n%s" % (knob, tail)
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


@pytest.mark.gpu
def test_attn_two_waves_per_head_forced_on_fewer_heads():
    """FIL_ATTN_WPH=2 forces the two-waves-per-head backward wherever there are more than 8 query blocks: test_attn_fused's
    (2, 230, 16, 2, 16) then runs it with TWO heads (wave w = head w mod 2, sub w / 2), fp32 -- the strict bar -- with and without the
    residual / LayerNorm branches; the shapes with up to 8 blocks keep one wave per head."""
    env = dict(os.environ)
    env["FIL_ATTN_WPH"] = "2"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py::test_attn_fused", "-x", "-q", "-m", "gpu", "-k", "230 or 200",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-15:])
    assert r.returncode == 0, "FIL_ATTN_WPH=2:\n%s" % tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail
