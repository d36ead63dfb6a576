"""C-ABI library: builds for gfx950, loads, exports every symbol include/rem2d.h declares, and its
host-only entry points behave (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd import _lib
    return _lib


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "rem2d.h")).read()
    names = sorted(set(re.findall(r"\b(rem2d_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 12
    for path in (lib.LIB_PATH, lib.WIDE_LIB_PATH):
        L = C.CDLL(path)
        for n in names:
            assert hasattr(L, n), "%s does not export %s" % (os.path.basename(path), n)


def test_field_table_matches_header(lib):
    hdr = open(os.path.join(ROOT, "include", "rem2d.h")).read()
    body = hdr[hdr.index("REM2D_F_PX = 0"):hdr.index("REM2D_F_COUNT")]
    ids = re.findall(r"\bREM2D_F_([A-Z0-9]+)\b", body)
    assert [i.lower() for i in ids] == lib.FIELDS


def test_error_bits_match_header(lib):
    """REM2D_ERR_* of include/rem2d.h are what gym_rem2d_amd._lib names (the step train's hand-over check added REM2D_ERR_HANDOVER
    in ABI 10: a creature that carries it is re-evaluated or refused like an overflowing one, never trusted)."""
    hdr = open(os.path.join(ROOT, "include", "rem2d.h")).read()
    bits = dict((k, int(v)) for k, v in re.findall(r"#define REM2D_ERR_([A-Z_]+) (\d+)", hdr))
    assert bits == {"PAIR_OVERFLOW": lib.ERR_PAIR_OVERFLOW, "SOLVER_OVERFLOW": lib.ERR_SOLVER_OVERFLOW, "HANDOVER": lib.ERR_HANDOVER}
    assert sorted(bits.values()) == [1, 2, 4]


def test_host_only_entry_points(lib):
    L = lib.lib()
    assert L.rem2d_abi_version() == 11
    assert lib.capacity() == (lib.CONTACT_SLOTS, lib.SOLVER_SLOTS) == (24, 6) and lib.capacity(wide=True) == (32, 12)
    # a wide world's arena is laid out for its own slot count
    big = lib.WorldCfg(4096, 8, 0, 0)
    assert lib.lib(wide=True).rem2d_state_bytes(C.byref(big)) > L.rem2d_state_bytes(C.byref(big))
    cfg = lib.WorldCfg(65536, 8, 0, 0)
    n = L.rem2d_state_bytes(C.byref(cfg))
    # every field is per lane / per slot / per creature: a few hundred bytes per body
    assert 400 * 65536 * 8 < n < 1200 * 65536 * 8
    assert L.rem2d_padded_envs(C.byref(cfg)) == 65536
    assert L.rem2d_padded_envs(C.byref(lib.WorldCfg(5, 4, 0, 0))) == 16
    assert L.rem2d_state_bytes(C.byref(lib.WorldCfg(10, 3, 0, 0))) == 0  # lanes must be a power of two
    h = C.c_void_p()
    rc = L.rem2d_world_create(C.byref(lib.WorldCfg(10, 3, 0, 0)), None, 0, C.byref(h))
    assert rc == -1 and b"lanes" in L.rem2d_last_error()
    assert L.rem2d_world_destroy(None) == 0
    assert L.rem2d_world_step(None, 1, None) == -1
    assert L.rem2d_groups_step(None, 0, 1, None, 0) == -1 and b"step groups" in L.rem2d_last_error()
    assert L.rem2d_world_set_option(None, 0, 3) == -1 and L.rem2d_world_get_option(None, 0, None) == -1
    # iteration counts the 16-bit tick counters of the solver loops cannot hold are refused, not truncated
    fake = C.c_void_p(8)   # (never dereferenced: the argument check comes first)
    assert L.rem2d_world_step_ex(fake, 1, 0.02, 70000, 60, None) == -1 and b"0..8192" in L.rem2d_last_error()
    assert L.rem2d_world_step_ex(fake, 1, 0.02, 180, -1, None) == -1


def test_library_reads_no_environment_variable():
    """The library's launch options are rem2d_world_set_option arguments (include/rem2d.h REM2D_OPT_*): the HIP sources
    must not call getenv, and the REM2D_* experiment overrides live in gym_rem2d_amd/_lib.py alone."""
    csrc = os.path.join(ROOT, "gym_rem2d_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    hdr = open(os.path.join(ROOT, "include", "rem2d.h")).read()
    ids = re.findall(r"\bREM2D_OPT_([A-Z0-9_]+)\b", hdr[hdr.index("REM2D_OPT_PIPELINE = 0"):hdr.index("REM2D_OPT_COUNT")])
    from gym_rem2d_amd import _lib
    assert [i.lower() for i in ids] == list(_lib.OPTIONS)


def test_product_never_imports_oracle():
    """The product path must not import, link, load or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "gym_rem2d_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|librem2d_oracle|oracle[/\\]|rem2d_oracle", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not bad.search(src), "%s reaches into the oracle" % f


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/librem2d.so")
    with pytest.raises(lib.Rem2dError):
        lib.lib()


def test_build_identity(lib, tmp_path, monkeypatch):
    """rem2d_build_id() is the hash of csrc/*, include/rem2d.h and the compile flags that build() compiled in; lib() recomputes
    it from the sources beside the library and refuses a library built from anything else (the .so files are git-ignored but
    travel to the GPU box: a stale one must not be tested or benched silently).  Here: a header is touched WITHOUT a rebuild."""
    import shutil
    for wide, flags, path in ((False, [], lib.LIB_PATH), (True, lib.WIDE_FLAGS, lib.WIDE_LIB_PATH)):
        assert lib.build_id(wide) == lib.source_id(flags) == lib.file_build_id(path)
        assert re.fullmatch(r"[0-9a-f]{16}", lib.build_id(wide))
    assert lib.build_id(False) != lib.build_id(True)          # (the flags are part of the identity)
    # a copy of the sources with one more comment line in a header, the library itself left alone
    csrc = tmp_path / "csrc"
    shutil.copytree(os.path.dirname(lib.SRC_PATH), str(csrc))
    inc = tmp_path / "include"
    inc.mkdir()
    shutil.copy(os.path.join(lib.INCLUDE_DIR, "rem2d.h"), str(inc / "rem2d.h"))
    with open(str(csrc / "rem2d_math.h"), "a") as f:
        f.write("// touched\n")
    monkeypatch.setattr(lib, "SRC_PATH", str(csrc / "rem2d.hip"))
    monkeypatch.setattr(lib, "INCLUDE_DIR", str(inc))
    monkeypatch.setattr(lib, "_lib", None)
    assert lib.source_id() != lib.file_build_id(lib.LIB_PATH)
    with pytest.raises(lib.Rem2dError, match="stale"):
        lib.lib()
    # an experiment's variant build named through REM2D_LIB_PATH is exempt (tools/build_variant.sh)
    monkeypatch.setenv("REM2D_LIB_PATH", lib.LIB_PATH)
    assert lib.lib().rem2d_abi_version() == 11
    monkeypatch.setattr(lib, "_lib", None)


def test_step_train_handover_sequences_in_the_code_objects(lib):
    """The step train's hand-over as COMPILED (tools/check_handover_asm.py; VERDICT r5 weak 2: the publish was ordered by the
    accident of a register spill).  In every build: walking back from the flag store every path meets `s_waitcnt vmcnt(0)` before
    any store to memory; after a poll of the flag every path meets buffer_inv sc1 + s_dcache_inv before any other load.  And the
    checker itself notices when they are missing: the same disassembly with the waits / the invalidate taken out is refused."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_handover_asm as H
    import tempfile
    for path in (lib.LIB_PATH, lib.WIDE_LIB_PATH, lib.FMA_LIB_PATH):
        rep = H.check_library(path)
        assert len(rep) >= 1
        for sym, r in rep.items():
            # (4 wavefronts per SIMD = 128 VGPRs; the static 128-lane shape's train is compiled for 3 = 168)
            assert r["polls"] >= 1 and r["publish_waitcnt"] and r["resources"]["vgprs"] <= (168 if "ILb0ELi3EE" in sym else 128), (path, sym, r)
        assert len(rep) == 3, "rem2d_step_train_kernel and the two rem2d_step_train128_kernel instantiations"
    # negative controls on the default build's kernel(s)
    with tempfile.TemporaryDirectory() as wd:
        co = H.code_object(lib.LIB_PATH, wd)
        for sym in H.train_symbols(co):
            ins = H.disassemble(co, sym)
            H.check_kernel(ins, sym)
            nop = lambda i: (i[0], "s_nop", "0", None)
            no_wait = [nop(i) if (i[1] == "s_waitcnt" and "vmcnt(0)" in i[2]) else i for i in ins]
            with pytest.raises(H.HandoverAsmError, match="s_waitcnt vmcnt|flag store"):
                H.check_kernel(no_wait, sym)
            no_inv = [nop(i) if i[1] in ("buffer_inv", "s_dcache_inv") else i for i in ins]
            with pytest.raises(H.HandoverAsmError):
                H.check_kernel(no_inv, sym)
            # only the scalar-cache invalidate missing (round 5's late fix b600711): refused as well
            no_dc = [nop(i) if i[1] == "s_dcache_inv" else i for i in ins]
            with pytest.raises(H.HandoverAsmError, match="s_dcache_inv"):
                H.check_kernel(no_dc, sym)
