"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and its host-only entry points (expression compile, argument validation, error
convention) behave.  No compute is launched (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from pairec_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "pairec_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == syms          # the ctypes binding lists exactly the header's API


def test_no_oracle_in_product():
    """The product path must not route through the oracle or any CPU fallback."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pairec_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_error_convention_without_gpu():
    L = _lib.load()
    h = C.c_void_p()
    rc = L.pg_init(0, None, C.byref(h))
    if rc == 0:                                   # running on a GPU box: nothing to check here
        L.pg_shutdown(h)
        pytest.skip("GPU present")
    assert rc < 0
    assert len(L.pg_last_error()) > 0            # thread-local message, no abort
    assert L.pg_init(0, None, None) == -1        # PG_ERR_INVALID


def test_expr_compile_host_side():
    L = _lib.load()

    def compile_(src):
        h = C.c_void_p()
        rc = L.pg_expr_compile(src.encode(), C.byref(h))
        if rc != 0:
            return rc, None
        names = [L.pg_expr_var_name(h, i).decode() for i in range(L.pg_expr_num_vars(h))]
        L.pg_expr_free(h)
        return rc, names

    assert compile_("${ctr} + ${click} + ${price}") == (0, ["ctr", "click", "price"])
    assert compile_("(${a}+2*${b})*${a}^0.1") == (0, ["a", "b"])       # variables are deduplicated
    assert compile_("-5") == (0, [])
    assert compile_("") == (0, [])                                      # GetExpAST("") → nil, no error
    assert compile_("abc")[0] == -6                                     # PG_ERR_PARSE ("symbol error")
    assert compile_("1 +\t")[0] == -6
    assert b"symbol error" in L.pg_last_error()


def test_null_argument_validation():
    L = _lib.load()
    assert L.pg_recall_topk(None, None, None, 1, 1, None, None, None) == -1
    assert L.pg_sort_scores(None, None, None, 0, 1, None) == -1
    assert L.pg_table_info(None, None, None, None) == -1
    assert b"NULL" in L.pg_last_error() or b"null" in L.pg_last_error().lower()
