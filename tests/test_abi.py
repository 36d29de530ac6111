"""The C ABI as a binary artefact: the library loads, exports every function
the header declares, fails loudly without a GPU, and the CLI keeps the
reference's usage / error contract.  No GPU needed (no compute call is made).
"""
import ctypes as C
import os
import re
import subprocess

import pytest

import orc
import raytracer_amd as rta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raytracer_amd.h")
PROBE_HEADER = os.path.join(ROOT, "include", "raytracer_amd_probe.h")


def declared_functions(header=HEADER):
    with open(header) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rta_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_function_is_exported():
    lib = C.CDLL(rta.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 24
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_binding_covers_the_header():
    assert set(rta.exported_symbols()) == set(declared_functions())


def test_probe_hooks_live_in_their_own_library():
    """Measurement and test hooks are declared in a header of their own and exported by the probe library only."""
    names = declared_functions(PROBE_HEADER)
    assert set(names) == set(rta.probe_symbols()) and names
    product, probe = C.CDLL(rta.LIB_PATH), C.CDLL(rta.PROBE_LIB_PATH)
    for name in names:
        assert hasattr(probe, name), f"{name} declared in the probe header but not exported"
        assert not hasattr(product, name), f"{name} is a probe hook but the product library exports it"


def test_struct_sizes_match_the_header(tmp_path):
    """The header compiles as plain C, and the ctypes mirrors have its sizes."""
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "raytracer_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(rta_scene_settings),sizeof(rta_scene_desc),sizeof(rta_camera),sizeof(rta_render_params),"
                   "sizeof(rta_stats),sizeof(rta_hit));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                    str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True).stdout.split()]
    mirrors = [rta.SceneSettings, rta.SceneDesc, rta.Camera, rta.RenderParams, rta.Stats]
    assert sizes[:5] == [C.sizeof(m) for m in mirrors]
    assert sizes[5] == rta.HIT_DTYPE.itemsize == 16


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert rta.device_count() == 0
    with pytest.raises(rta.RtaError) as e:
        rta.Context(0)
    assert e.value.code == rta.E_NO_DEVICE
    null = C.c_void_p()
    assert rta.lib().rta_render(null, None, None, None, None, None) == rta.E_INVALID
    assert rta.lib().rta_trace_closest(null, None, 0, None, 0) == rta.E_INVALID


def test_product_does_not_link_the_checker():
    out = subprocess.run(["ldd", rta.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out and "libref" not in out
    with open(rta.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"orc_render" not in blob and b"liboracle" not in blob


def test_cli_usage_and_errors():
    p = subprocess.run([rta.CLI_PATH], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 1 and p.stderr.startswith("Usage: ") and p.stderr.strip().endswith("[FILE]")
    p = subprocess.run([rta.CLI_PATH, "/nonexistent.txt"], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 1 and "Couldn't open '/nonexistent.txt'" in p.stderr
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "bad.txt"), "w") as f:
            f.write("render { widht 4 }\n")
        p = subprocess.run([rta.CLI_PATH, "bad.txt"], cwd=tmp, stderr=subprocess.PIPE, text=True)
        assert p.returncode == 1 and p.stderr.strip() == "ERROR:bad.txt[10]: Unknown key"
