"""The C-ABI shared library loads on a CPU-only machine (no HIP call at load time) and
exports every symbol include/cocons_hip.h declares.  No compute calls here."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cocons_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cocons_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from cocons_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(lib, nm), "missing export " + nm
        assert nm in _lib.SIGNATURES, "python binding lacks " + nm
    assert sorted(_lib.SIGNATURES) == names
    assert lib.cocons_abi_version() == 1


def test_no_oracle_or_torch_in_product_library():
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "cocons_amd", "csrc", "libcocons_hip.so")]).decode()
    assert "oracle" not in out and "torch" not in out
    assert "amdhip64" in out


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cocons_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace(
                    "nothing from oracle/", ""), fn


def test_host_penalty_runs_without_gpu():
    """cocons_sumsmoothlone is pure host arithmetic (src/cocons_full.cpp:12-30)."""
    import cocons_amd as ca
    assert ca.sumsmoothlone([0.5, -2.0], 2.0) == 5.0
