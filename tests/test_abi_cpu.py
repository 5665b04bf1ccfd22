"""CPU-side checks of the C-ABI library: it loads, exports exactly what
include/diffus_hip.h declares, and validates arguments before touching HIP.
No compute is launched here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from diffus_amd import build, _lib
    build.build()
    return _lib.load()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "diffus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(diffus_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_exports_agree(lib):
    from diffus_amd import _lib
    names = header_functions()
    assert len(names) >= 10
    assert sorted(_lib.EXPORTS) == names
    for n in names:
        assert getattr(lib, n) is not None          # dlsym succeeds
    assert lib.diffus_abi_version() == _lib.ABI_VERSION == 8
    assert lib.diffus_strerror(0) == b"ok"
    assert b"workspace" in lib.diffus_strerror(-4)


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "diffus_hip.h")).read()
    for fn in ("diffus_render_fwd", "diffus_render_bwd", "diffus_trace_rays", "diffus_echo_traces"):
        i = txt.index("int " + fn)
        comment = txt[txt.rfind("/*", 0, i): i]
        assert "src/renderer.py:" in comment, fn


def test_sizes_are_host_side_arithmetic(lib):
    assert lib.diffus_bricked_floats(256, 256, 256) == 256 ** 3
    assert lib.diffus_bricked_floats(5, 7, 3) == 2 * 2 * 2 * 32
    assert lib.diffus_bricked_floats(0, 4, 4) == 0
    w = lib.diffus_workspace_bytes(32, 256, 512, 0)
    assert w >= 32 * 256 * 512 * 4 and w % 256 == 0
    assert lib.diffus_workspace_bytes(0, 1, 1, 0) == 0


def test_argument_validation_without_a_gpu(lib):
    f = (C.c_float * 8)()
    p = C.cast(f, C.c_void_p)
    ok = dict(vol=p, d0=2, d1=2, d2=2, layout=0, src=p, sdt=0, dirs=p, ddt=0, P=1, R=1, S=4, start=0, alpha=0.1, sampler=0)

    def fwd(**kw):
        a = dict(ok, **kw)
        return lib.diffus_render_fwd(a["vol"], a["d0"], a["d1"], a["d2"], a["layout"], a["src"], a["sdt"], a["dirs"],
                                     a["ddt"], a["P"], a["R"], a["S"], a["start"], a["alpha"], a["sampler"],
                                     kw.get("frame", p), None, None, 0, None)

    assert fwd(vol=None) == -1
    assert fwd(d0=0) == -1
    assert fwd(sampler=7) == -1
    assert fwd(layout=3) == -1
    assert fwd(sdt=3) == -1
    assert fwd(start=4) == -1            # start > S-1
    assert fwd(start=3) == -1            # start > 0 needs two samples (reference IndexError at :243)
    assert fwd(S=70000) == -2            # S - start > DIFFUS_MAX_SAMPLES * DIFFUS_MAX_SEGMENTS
    assert fwd(S=2000) == -4             # long rays need the workspace for their carries
    assert fwd(d0=1 << 25) == -2
    assert fwd(start=1) == -4            # start > 0 needs the workspace
    assert fwd(frame=None) == -1
    assert lib.diffus_render_bwd(p, 2, 2, 2, 0, p, 0, p, 0, 1, 1, 4, 0, 0.1, 0, p, p, None, None, None, 3, None, 0, None) == -4
    assert lib.diffus_render_bwd(p, 2, 2, 2, 0, p, 0, p, 0, 1, 1, 4, 0, 0.1, 0, p, p, None, None, None, 0, p, 1 << 20, None) == -1
    assert lib.diffus_render_bwd(p, 2, 2, 2, 0, p, 0, p, 0, 1, 1, 4, 0, 0.1, 0, p, None, None, None, None, 3, None, 0, None) == 0
    assert lib.diffus_brick_count(256, 256, 256) == 256 ** 3 // 32
    assert lib.diffus_gradbuf_flush(None, p, 2, 2, 2, p, 1, None) == -1
    assert lib.diffus_echo_traces(None, 1, 4, p, None) == -1
    assert lib.diffus_brick_volume(None, 2, 2, 2, p, None) == -1
    assert lib.diffus_convert_volume_box(None, 2, 2, 2, 1, p, 0, 1, 0, 1, 0, 1, None) == -1      # null volume
    assert lib.diffus_convert_volume_box(p, 2, 2, 2, 0, p, 0, 1, 0, 1, 0, 1, None) == -1         # canonical is no converted layout
    assert lib.diffus_convert_volume_box(p, 2, 2, 2, 2, p, 0, 3, 0, 1, 0, 1, None) == -1         # box outside the volume
    assert lib.diffus_convert_volume_box(p, 2, 2, 2, 2, p, 1, 1, 0, 2, 0, 2, None) == 0          # empty box: nothing launched
    assert lib.diffus_loss_sumsq(p, 0, 4, p, None, p, 1024, None) == -1
    assert lib.diffus_loss_sumsq(p, 1, 4, p, None, None, 0, None) == -4


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import diffus_amd
    from diffus_amd import _lib
    if not torch.cuda.is_available():
        with pytest.raises(diffus_amd.DiffusError, match="no CPU fallback"):
            diffus_amd.UltrasoundRenderer(8).plot_beam_frame(torch.ones(4, 4, 4), torch.zeros(3), torch.ones(2, 3))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(diffus_amd.DiffusError, match="not found"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffus_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "liboracle" not in src, fn
