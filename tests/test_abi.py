"""CPU: the C-ABI library is present, loads, and exports every symbol include/gpbo.h declares
(no compute call is made here: there is no GPU in this environment)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'gpbo.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gpbo_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ('gpbo_create', 'gpbo_destroy', 'gpbo_fit', 'gpbo_set_candidates', 'gpbo_posterior', 'gpbo_acquire',
                 'gpbo_predict_ei', 'gpbo_read', 'gpbo_append', 'gpbo_lml_batch', 'gpbo_get_factor', 'gpbo_last_timings'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from bayesian_optimization_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'libgpbo.so not built: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_python_binding_covers_the_header():
    from bayesian_optimization_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_missing_library_fails_loudly(monkeypatch):
    from bayesian_optimization_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', os.path.join(ROOT, 'no_such_dir', 'libgpbo.so'))
    try:
        _lib.load()
    except ImportError as e:
        assert 'no CPU' in str(e) or 'not found' in str(e)
    else:
        raise AssertionError('loading a missing library must raise')


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'bayesian_optimization_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.lower().replace('the oracle check', ''), os.path.join(dirpath, f)
