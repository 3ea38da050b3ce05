"""C++ facade (include/xq/xq.hpp — ChessBoard / DQN / ChessAI / ReplayBuffer / VecEnv over the C ABI).

not gpu: the header compiles with plain g++ against include/xq_capi.h (no HIP headers needed by a consumer).
gpu:     tests/cpp/facade_test.cpp runs the facade against the CPU oracle on the device."""
import os
import subprocess

import pytest

import xqoracle as xo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "facade_test")


def build():
    xo.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    pkg = os.path.join(ROOT, "cn_chess_ai_amd")
    orc = os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", SRC, "-o", BIN, f"-L{pkg}", "-lxqhip", f"-L{orc}",
                           "-lxqoracle", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}", "-Wl,-rpath,/opt/rocm/lib"])


def test_facade_compiles_with_plain_gxx():
    if not os.path.exists(os.path.join(ROOT, "cn_chess_ai_amd", "libxqhip.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cn_chess_ai_amd", "csrc"), "all"])
    build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_facade_against_oracle_on_device(tmp_path):
    build()
    out = subprocess.run([BIN], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "all checks passed" in out.stdout


EXAMPLE = os.path.join(ROOT, "examples", "train_selfplay.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "tests", "cpp", "_build", "train_selfplay")


def build_example():
    os.makedirs(os.path.dirname(EXAMPLE_BIN), exist_ok=True)
    pkg = os.path.join(ROOT, "cn_chess_ai_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", EXAMPLE, "-o", EXAMPLE_BIN, f"-I{os.path.join(ROOT, 'include')}",
                           f"-L{pkg}", "-lxqhip", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"])


def test_example_compiles():
    build_example()


@pytest.mark.gpu
def test_example_trains_and_saves_a_reference_format_model(tmp_path):
    build_example()
    model = tmp_path / "m.bin"
    out = subprocess.run([EXAMPLE_BIN, "300", str(model), "256"], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0 and "300 episodes" in out.stdout
    assert model.stat().st_size == 9650484                      # DQN::saveModel layout for {1260,128,8100} (dqn.cpp:76-108)
