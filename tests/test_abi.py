"""CPU: the C-ABI library is present, loads, and exports every symbol include/gpbo.h declares
(no compute call is made here: there is no GPU in this environment)."""
import ctypes
import os
import re

import pytest

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


def test_header_is_plain_c99_and_the_c_caller_links():
    """include/gpbo.h must be usable from C (no C++ constructs): the header compiles under gcc -std=c99 -pedantic, and the
    C99 caller of the GPU tests builds and links against the library (no GPU call is made here)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = os.path.join(ROOT, 'tests', 'c_caller', 'bo_body.c')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), src],
                   check=True, timeout=120)
    import __graft_entry__ as ge
    ge.build()
    caller = os.path.join(ROOT, 'tests', 'c_caller', 'bo_body')
    assert os.path.exists(caller)
    und = subprocess.run(['nm', '-D', '--undefined-only', caller], check=True, stdout=subprocess.PIPE, timeout=60).stdout.decode()
    for sym in ('gpbo_create', 'gpbo_fit', 'gpbo_predict_ei', 'gpbo_lml_batch', 'gpbo_destroy', 'gpbo_last_error'):
        assert sym in und
