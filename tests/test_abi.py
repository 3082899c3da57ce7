"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol that include/neube_hip.h declares (no kernels are launched here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from brushstroke_engine_amd import _lib, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def library():
    build.build()          # no-op when up to date; hipcc cross-compiles without a GPU
    return _lib.lib()


def declared_symbols(header="neube_hip.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(library):
    names = declared_symbols()
    assert len(names) >= 12
    assert sorted(_lib.PROTOTYPES) == names
    for n in names:
        assert hasattr(library, n), n


def test_debug_hooks_are_declared_apart(library):
    """Everything the library exports is declared: the product ABI in neube_hip.h, the process-global developer / test
    switches in neube_hip_debug.h (none of which the product path calls)."""
    import subprocess
    dbg = declared_symbols("neube_hip_debug.h")
    assert dbg and not set(dbg) & set(declared_symbols())
    for n in dbg:
        assert hasattr(library, n), n
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\bT (nb_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == sorted(set(dbg) | set(declared_symbols())), sorted(set(exported) ^ (set(dbg) | set(declared_symbols())))
    pkg = os.path.join(REPO, "brushstroke_engine_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "nb_debug_" not in open(os.path.join(root, f)).read(), f


def test_abi_version_and_error_string(library):
    assert library.nb_abi_version() == _lib.ABI_VERSION
    # argument validation happens on the host before any launch: usable without a GPU
    rc = library.nb_bias_act_f32(None, None, None, 4, 0, 1, 3, 0.2, 1.0, -1.0, None)
    assert rc == -1
    assert b"null pointer" in library.nb_last_error()
    with pytest.raises(_lib.NeubeHipError):
        _lib.check(rc, "bias_act")


def test_layer_desc_layout_matches_header(library):
    # 9 pointers + 5 int32 + 1 float + 2 pad int32 = 72 + 32 = 104 bytes, 8-byte aligned
    assert ctypes.sizeof(_lib.NbLayerDesc) == 104


def test_pack_conv_weight_host_helper(library):
    rs = np.random.RandomState(0)
    w = rs.randn(8, 5, 3, 3).astype(np.float32)
    wpk = np.full((8, 9, 32), 7.0, np.float32)        # padded to [ceil8(c_in)][9][ceil32(c_out)]
    wsq = np.zeros((5, 8), np.float32)
    rc = library.nb_pack_conv_weight(w.ctypes.data, 8, 5, wpk.ctypes.data, wsq.ctypes.data)
    assert rc == 0
    np.testing.assert_array_equal(wpk[:5, :, :8], w.transpose(1, 2, 3, 0).reshape(5, 9, 8))
    assert float(np.abs(wpk[5:]).max()) == 0.0 and float(np.abs(wpk[:, :, 8:]).max()) == 0.0
    np.testing.assert_allclose(wsq, (w ** 2).sum(axis=(2, 3)).T, rtol=1e-6)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No library and no compiler: every product-path op raises (there is no CPU fallback)."""
    from brushstroke_engine_amd import build
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(build, "LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(build, "STAMP", str(tmp_path / "nope.so.stamp"))
    monkeypatch.setattr(build, "HIPCC", str(tmp_path / "no-hipcc"))
    with pytest.raises(_lib.NeubeHipError, match="no CPU fallback"):
        _lib.lib()


def test_stale_library_is_detected(monkeypatch, tmp_path):
    """A library built from other sources than the tree holds (a kernel edit without a rebuild) is not loaded silently:
    the content digest in the stamp file no longer matches."""
    from brushstroke_engine_amd import build
    assert not build.is_stale()                                   # conftest / build() left a current library
    stamp = tmp_path / "stamp"
    stamp.write_text("0" * 64 + "\n")
    monkeypatch.setattr(build, "STAMP", str(stamp))
    assert build.is_stale()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("NEUBE_NO_AUTOBUILD", "1")
    with pytest.raises(_lib.NeubeHipError, match="built from different sources"):
        _lib.lib()


def test_stampless_prebuilt_library_loads_with_warning(monkeypatch, tmp_path):
    """A prebuilt library WITHOUT a build stamp (packaged deployment, hand build) is loaded -- with a warning, and still
    subject to the export / ABI-version checks -- instead of failing on machines without hipcc or rebuilding silently."""
    from brushstroke_engine_amd import build
    assert os.path.exists(build.LIB)
    monkeypatch.setattr(build, "STAMP", str(tmp_path / "absent.stamp"))
    monkeypatch.setattr(build, "HIPCC", str(tmp_path / "no-hipcc"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.warns(UserWarning, match="no build stamp"):
        l = _lib.lib()
    assert l.nb_abi_version() == _lib.ABI_VERSION


def test_no_scratch_in_counted_wait_kernels(library):
    """The split-f16 and fp32-MFMA conv kernels keep several LDS-DMA operations in flight and wait on COUNTS
    (``s_waitcnt vmcnt(N)``); register spills or other compiler-made scratch accesses inside their K loops would sit in the same
    counter.  Two checks on the built library: (1) the code-object metadata shows no scratch at all -- except for the PERSISTENT
    kernels of round 6 (a tile loop around the K loop: the 8-wave up=1 kernel and up2v), which may hold a few registers in scratch
    AROUND the K loop (spilled before a tile's loop, reloaded behind it) -- and (2) the disassembly of EVERY kernel shows no scratch
    instruction and no SGPR-spill lane traffic inside any matrix loop (innermost loops with >= 8 MFMA instructions)."""
    res = build.kernel_resources()
    assert len(res) >= 60
    known = {"modconv3x3_up1_small_h3_kernel"}          # no counted waits: weight fragments by plain loads, vmcnt(0) only
    persistent = ("modconv3x3_up1_h3_kernel", "modconv3x3_up2v_kernel")
    bad = {k: v for k, v in res.items() if (v["scratch_bytes"] or v["vgpr_spill"]) and not any(n in k for n in known)
           and not (any(n in k for n in persistent) and v["vgpr_spill"] <= 40)}
    assert not bad, bad
    for k, v in res.items():
        assert v["vgpr"] <= 256 or "up2_h3" not in k, (k, v)
    loops = build.mfma_loop_spill_traffic()
    assert sum(len(v) for v in loops.values()) >= 40, "the disassembly walk found no matrix loops"
    # (lane traffic: one v_readlane per six-step body of the round-3 H2 loop has been there since round 3)
    hot = {k: v for k, v in loops.items() if any(sc or lanes > 2 for _, sc, lanes in v) and not any(n in k for n in known)}
    assert not hot, hot
    # ... and the walk sees the loops it is meant to guard: the steady-state bodies of the two persistent kernels
    for name, n_mfma in (("modconv3x3_up1_h3_kernelILi2ELb1ELi2ELb1ELb0ELb1ELb0E", 108), ("modconv3x3_up2v_kernelILb1ELi2ELb0ELb0E", 28)):
        rows = [v for k, v in loops.items() if name in k]
        assert rows and any(r[0] == n_mfma for r in rows[0]), (name, rows)


def test_no_packed_f32_high_dword_broadcast(library):
    """`v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]` with the default op_sel_hi (both results read the HIGH dword of a register
    pair) was the one instruction whose removal ended run-to-run differences of the up=2 epilogue on MI355X (its noise
    add; a stand-alone loop of the instruction does not reproduce it, so this is a tripwire, not a diagnosis).  No kernel of the library may contain that operand form."""
    dis = build.disassembly()
    assert dis.count("v_mfma_") > 100                             # (it is the real disassembly)
    bad = [ln.strip() for ln in dis.splitlines() if re.search(r"v_pk_(add|mul|fma)_f32 .*op_sel:\[0,1\](?! op_sel_hi)", ln)]
    assert not bad, bad[:5]
    # ... and, more generally, no packed fp32 instruction that SWIZZLES register halves: the only operand forms the kernels are
    # written to use are the plain one (no op_sel) and the low-dword broadcast `op_sel_hi:[..]` with zeros (a scalar against a
    # pair).  Forms such as `op_sel:[0,1] op_sel_hi:[1,0]` / `op_sel:[1,0] op_sel_hi:[0,1]` come from the SLP vectoriser pairing
    # scalar arithmetic and returned wrong halves sporadically on MI355X (csrc/nb_common.h, NB_NO_PACKED_F32).
    def swizzled(ln):
        m = re.search(r"v_pk_(?:add|mul|fma)_f32 (.*)", ln)
        if not m:
            return False
        sel = re.search(r"op_sel:\[([01,]+)\]", ln)
        return bool(sel and "1" in sel.group(1))          # any op_sel bit set = a low result lane reads a HIGH dword
    bad = [ln.strip() for ln in dis.splitlines() if swizzled(ln)]
    assert not bad, bad[:5]
