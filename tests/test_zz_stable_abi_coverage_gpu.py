"""Runs LAST in a GPU session (file name): every symbol of the STABLE C ABI (include/cppf_hip.h = cppf2_amd._lib.STABLE) must have been
called through ctypes by some GPU test of this session -- an entry point nothing exercises on hardware is not part of a stable
interface.  tests/conftest.py sets CPPF_ABI_TRACE, which makes cppf2_amd._lib.load() record every call's symbol in _lib.CALLED."""
import os

import pytest


@pytest.mark.gpu
def test_every_stable_symbol_was_called_on_the_gpu(request):
    from cppf2_amd import _lib
    files = {os.path.basename(str(it.fspath)) for it in request.session.items}
    if len(files) < 12:
        pytest.skip("not a full GPU session (%d test files collected): coverage is a property of the whole suite" % len(files))
    assert os.environ.get("CPPF_ABI_TRACE") and isinstance(_lib.load(), _lib._Traced)
    missing = sorted(set(_lib.STABLE) - _lib.CALLED)
    assert not missing, "stable C-ABI symbols no GPU test called: %s" % missing
    # informational: the experimental ones that were exercised too
    print("stable %d/%d called; experimental %d/%d" % (len(set(_lib.STABLE) & _lib.CALLED), len(_lib.STABLE),
                                                      len(set(_lib.EXPERIMENTAL) & _lib.CALLED), len(_lib.EXPERIMENTAL)))


def test_call_tracing_wraps_every_symbol_and_costs_nothing_when_off(monkeypatch):
    """CPU: the tracer forwards calls and records names; without CPPF_ABI_TRACE load() hands out the CDLL itself."""
    from cppf2_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("CPPF_ABI_TRACE", "1")
    lib = _lib.load()
    assert isinstance(lib, _lib._Traced)
    _lib.CALLED.discard("cppf_version")
    assert lib.cppf_version() == _lib.ABI_VERSION and "cppf_version" in _lib.CALLED
    assert lib.cppf_shot352_workspace_bytes(1, 100) > 0 and "cppf_shot352_workspace_bytes" in _lib.CALLED
    with pytest.raises(AttributeError):
        lib.cppf_no_such_symbol
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("CPPF_ABI_TRACE")
    import ctypes
    assert isinstance(_lib.load(), ctypes.CDLL)
    monkeypatch.setattr(_lib, "_lib", None)          # (the session's traced object is rebuilt on the next load)
