"""The C-ABI library loads on a CPU-only machine and exports every function include/slimm_hip.h declares."""
import os
import re

from slimm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "slimm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(slimm_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared()
    assert len(declared) >= 30
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in slimm_hip.h but not exported by libslimm_hip.so"
    bound = {n for n, _, _ in capi.SYMBOLS}
    assert declared == bound, f"binding out of sync: {declared ^ bound}"


def test_version_and_no_gpu_failure_is_loud():
    L = capi.lib()
    assert b"gfx950" in L.slimm_version()
