"""CPU-only checks of the boundary (not gpu): the C-ABI library builds for gfx950 without a GPU, loads, exports every
symbol include/xq_capi.h declares, and FAILS LOUDLY instead of falling back when no device is present."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xq_capi.h")


@pytest.fixture(scope="module")
def capi():
    lib = os.path.join(ROOT, "cn_chess_ai_amd", "libxqhip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cn_chess_ai_amd", "csrc"), "all"])
    from cn_chess_ai_amd import _capi
    _capi.load()
    return _capi


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xq_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(capi):
    names = declared_functions()
    assert len(names) >= 55
    lib = capi.load()
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (xq_[a-z0-9_]+)", out))
    for n in names:
        assert n in exported, f"{n} declared in xq_capi.h but not exported"
        assert n in capi.PROTOTYPES, f"{n} has no ctypes prototype"
        getattr(lib, n)
    assert set(capi.PROTOTYPES) == set(names)


def test_header_cites_reference_for_each_entry_point():
    text = open(HEADER).read()
    for ref in ("chessboard.cpp", "chessai.cpp", "dqn.cpp", "dqn.cu", "dqn.h", "chessboard.h", "action.h"):
        assert ref in text


def test_library_targets_gfx950_only(capi):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={capi.LIB_PATH}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_struct_layouts_match_header(capi):
    assert C.sizeof(capi.StepResult) == 24 and C.sizeof(capi.EpisodeRecord) == 16
    assert capi.StepResult.move_count.offset == 18 and capi.StepResult.red_score.offset == 20


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU box: the no-device error path cannot be exercised")
def test_no_device_fails_loudly(capi):
    import cn_chess_ai_amd as xq
    assert capi.device_count() == 0
    from cn_chess_ai_amd import dist as xd
    for make in (lambda: xq.VecEnv(4), lambda: xq.DQN([1260, 128, 8100]), lambda: xq.ReplayBuffer(16),
                 lambda: xq.Trainer(xq.TrainerConfig(n_games=4)), lambda: xd.Comm(rank=0, world=1, id_bytes=bytes(128))):
        with pytest.raises(xq.XqError) as e:
            make()
        assert e.value.code == 3 and "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cn_chess_ai_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "xqoracle" not in text and "xq_oracle" not in text and "oracle/" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "oracle" not in open(p).read()


def test_host_constants_agree_with_oracle(capi):
    import cn_chess_ai_amd as xq
    import xqoracle as xo
    assert np.array_equal(xq.START_BOARD, xo.new_board().squares())
    assert xq.eps_to_u32(0.1) == xo.eps_to_u32(0.1) == 429496729
    assert xq.eps_to_u32(0.0) == 0 and xq.eps_to_u32(1.0) == 4294967295 and xq.eps_to_u32(-3) == 0
