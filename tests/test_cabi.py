"""The C-ABI library must load on a machine without a GPU and export every symbol include/rrtmg_lw_hip.h declares;
without a device every entry point must fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rrtmg_lw_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rrtmg_lw_hip_\w+)\s*\(", src)))


def test_header_symbols_are_exported():
    from rrtmg_lw_amd import api
    lib = api.lib()
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rrtmg_lw_hip.h but not exported"


def test_g256_library_exports_the_same_symbols():
    """librrtmg_lw_hip_g256.so is the same translation unit compiled with -DRRLW_G256: same C ABI."""
    from rrtmg_lw_amd import api
    assert os.path.exists(api.LIB_PATH_G256), "build it with __graft_entry__.build()"
    lib = ctypes.CDLL(api.LIB_PATH_G256)
    for n in _declared():
        assert hasattr(lib, n), n
    assert lib.rrtmg_lw_hip_gpoints() == 256
    assert api.lib().rrtmg_lw_hip_gpoints() == 140


def test_shipped_libraries_are_built_without_tuning_switches():
    """kernels.hip carries one-source measurement switches (-DRRLW_...: kernel geometry, numerics variants, two knock-outs that give wrong
    results).  The shipped libraries are built with none of them: the build-flags word says 0 (8 = the 256-g-point configuration) and the
    recorded compile command (<lib>.buildinfo, second line) holds no -DRRLW_ but -DRRLW_G256; a knock-out does not even compile outside a
    tuning build, and rrtmg_lw_hip_init refuses a library whose results are not the product's (checked on the source text: building one
    takes minutes)."""
    from rrtmg_lw_amd import api
    for path, want, allowed in ((api.LIB_PATH, 0, set()), (api.LIB_PATH_G256, 8, {"-DRRLW_G256"})):
        lib = ctypes.CDLL(path)
        lib.rrtmg_lw_hip_build_flags.restype = ctypes.c_uint
        assert lib.rrtmg_lw_hip_build_flags() == want, path
        info = path + ".buildinfo"
        if os.path.exists(info):
            lines = open(info).read().split("\n")
            if len(lines) > 1 and lines[1].strip():
                assert set(re.findall(r"-DRRLW_\w+", lines[1])) == allowed, lines[1]
    k = open(os.path.join(ROOT, "rrtmg_lw_amd", "csrc", "kernels.hip")).read()
    assert re.search(r"#if \(defined\(RRLW_KO_\w+\)( \|\| defined\(RRLW_KO_\w+\))*\) && !defined\(RRLW_TUNE\)\s*\n#error", k)
    guard = k[:k.index("#error")]
    for ko in set(re.findall(r"RRLW_KO_\w+", k)):
        assert ko in guard, f"{ko} is not fenced"
    d = open(os.path.join(ROOT, "rrtmg_lw_amd", "csrc", "driver.hip")).read()
    assert "RRTMG_LW_ALLOW_TUNE_BUILD" in d and "RRLW_BF_KNOCKOUT | RRLW_BF_NUMERICS" in d
    # every -D switch the sources test for is either a product configuration or named in the build-flags section
    used = set(re.findall(r"\bRRLW_[A-Z0-9_]+\b", re.sub(r"//.*", "", k) + re.sub(r"//.*", "", d)))
    used = {u for u in used if not u.startswith("RRLW_BF_")} - set(re.findall(r"#define (RRLW_\w+)\(", k + d))     # (function-like macros are not switches)
    head = k[:k.index("namespace rrlw {")]
    missing = sorted(u for u in used if u not in head)
    assert not missing, f"switches not covered by the build-flags word: {missing}"


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rrtmg_lw_amd import api
    with pytest.raises(api.RrtmgLwError, match="no HIP device|ENODEVICE|error 3"):
        api.rrtmg_lw_ini(1004.0, device=0)
    lib = api.lib()
    lib.rrtmg_lw_hip_last_error.restype = ctypes.c_char_p
    assert lib.rrtmg_lw_hip_kdata_is_standin() == -1
    assert lib.rrtmg_lw_hip_check(None) == 4          # RRTMG_LW_HIP_ENOTINIT
    # the host-pointer entry - a small call goes through the combining entry for concurrent callers, a large one straight to the lock -
    # says the same instead of touching a device that is not there
    from rrtmg_lw_amd.synth import make_gcm_inputs
    for ncol in (8, 9000):
        with pytest.raises(api.RrtmgLwError, match="rrtmg_lw_hip_init has not been called"):
            api.rrtmg_lw_from_dict(make_gcm_inputs(ncol, 20, "cloudy"))
    assert api.combine_stats()[0] >= 1
    with pytest.raises(api.RrtmgLwError, match="has not been called"):
        api.set_cu_partition(96)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, load or link oracle/."""
    pkg = os.path.join(ROOT, "rrtmg_lw_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|liboracle|libref_|#include\s+\"[^\"]*oracle", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".f90", ".h")):
                assert not bad.search(open(os.path.join(dp, f), errors="ignore").read()), (dp, f)


def test_output_arrays_are_checked_not_converted():
    """An array the C entries write through must be float64, Fortran-ordered and exactly shaped: a float32 / C-ordered / sliced array is
    refused before any pointer leaves Python (np.asfortranarray on an OUTPUT hands a converted copy to nobody)."""
    import numpy as np
    from rrtmg_lw_amd import api
    ncol, nlay, ng = 6, 5, api.gpoints()
    ok = np.zeros((ng, ncol, nlay), order="F")
    assert api._out_ok(ok, (ng, ncol, nlay), "x") is ok
    for bad in (np.zeros((ng, ncol, nlay), order="C"), np.zeros((ng, ncol, nlay), dtype=np.float32, order="F"),
                np.zeros((ng, ncol + 1, nlay), order="F"), np.zeros((ng, 2 * ncol, nlay), order="F")[:, ::2, :], [[0.0]]):
        with pytest.raises(ValueError, match="output array"):
            api._out_ok(bad, (ng, ncol, nlay), "x")
    z2 = lambda: np.zeros((ncol, nlay), order="F")
    out = dict(cldfmcl=ok, ciwpmcl=ok.copy(order="F"), clwpmcl=ok.copy(order="F"), taucmcl=ok.copy(order="F"), reicmcl=z2(),
               relqmcl=np.zeros((ncol, nlay), dtype=np.float32, order="F"))
    with pytest.raises(ValueError, match="relqmcl"):
        api.mcica_subcol_lw(ncol, nlay, 2, 1, 0, z2(), z2(), z2(), z2(), z2(), z2(), np.zeros((16, ncol, nlay), order="F"), out=out)
    o = {k: np.zeros((ncol, nlay + 1), order="F") for k in ("uflx", "dflx", "uflxc", "dflxc")}
    o["hr"] = np.zeros((ncol, nlay), order="C")
    o["hrc"] = z2()
    from rrtmg_lw_amd.synth import make_gcm_inputs
    with pytest.raises(ValueError, match="'hr'"):
        api.rrtmg_lw_from_dict(make_gcm_inputs(ncol, nlay, "clear"), out=o)


def test_shipped_code_objects_keep_every_exec_restore():
    """The libraries are compiled with -mllvm -amdgpu-remove-redundant-endcf=0 (__graft_entry__.CODEGEN_FLAGS): without it LLVM merges the end
    of an inner `if` with the end of the enclosing divergent region - the inner `if` then opens with a plain `s_and_b64 exec, exec, sN` -
    and copies the register allocator places behind the inner `if` run under its mask; k_layer's threads past the last column lost
    registers that way and whole waves of a ragged last window came out wrong (profiles/round5_exec_hazard.md).  Checked on what ships:
    the gfx950 code object inside each library, disassembled, holds no such narrowing; and the recorded compile command has the flag."""
    import shutil, subprocess, tempfile
    from rrtmg_lw_amd import api
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    for path in (api.LIB_PATH, api.LIB_PATH_G256):
        info = path + ".buildinfo"
        if os.path.exists(info):
            lines = open(info).read().split("\n")
            if len(lines) > 1 and lines[1].strip():
                assert "-amdgpu-remove-redundant-endcf=0" in lines[1], lines[1]
        tmp = tempfile.mkdtemp()
        try:
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
            subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", path, os.path.join(tmp, "unused.so")], check=True, capture_output=True)
            subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                           check=True, capture_output=True)
            asm = subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        assert asm.count("s_and_saveexec_b64") > 1000, "disassembly looks empty"
        n = len(re.findall(r"\bs_and_b64 exec, exec, s\[", asm))
        assert n == 0, f"{os.path.basename(path)}: {n} inner branches whose exec restore was merged away - built without {' '.join(('-mllvm', '-amdgpu-remove-redundant-endcf=0'))}?"


def test_exec_hazard_scanner_recognises_the_construct(tmp_path):
    """tools/exec_hazard.py on two hand-made listings: the merged end-cf of profiles/round5_exec_hazard.md (an inner `if` opened with a plain
    s_and_b64 on exec, the register allocator's copies behind it) is counted, the same code with its own saveexec / restore is not."""
    import subprocess, sys
    bad = """_Z1kv:
	s_and_saveexec_b64 s[28:29], vcc
	s_cbranch_execz .LBB0_5
	v_mov_b32_e32 v38, v162
	v_add_f64 v[162:163], v[56:57], v[116:117]
	s_and_b64 exec, exec, s[0:1]
	s_cbranch_execz .LBB0_4
	global_store_dwordx4 v[164:165], v[160:163], off nt
.LBB0_4:
	v_mov_b32_e32 v162, v38
.LBB0_5:
	s_or_b64 exec, exec, s[28:29]
	s_endpgm
"""
    good = bad.replace("\ts_and_b64 exec, exec, s[0:1]\n", "\ts_and_saveexec_b64 s[2:3], s[0:1]\n").replace(".LBB0_4:\n", ".LBB0_4:\n\ts_or_b64 exec, exec, s[2:3]\n")
    (tmp_path / "bad.s").write_text(bad)
    (tmp_path / "good.s").write_text(good)
    tool = os.path.join(ROOT, "tools", "exec_hazard.py")
    r = subprocess.run([sys.executable, tool, str(tmp_path / "bad.s")], capture_output=True, text=True)
    assert r.returncode == 1 and "1 plain exec narrowings, 1 with vector writes" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, tool, str(tmp_path / "good.s")], capture_output=True, text=True)
    assert r.returncode == 0 and "0 plain exec narrowings" in r.stdout, r.stdout
