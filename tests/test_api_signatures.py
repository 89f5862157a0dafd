"""Drop-in check of the host-side mirror: every function the reference exports exists here with the same
parameter names, order and literal defaults (tests/golden/api_signatures.json, taken from the reference's
source text by tests/golden/make_signatures.py).  Extra keyword parameters after the reference's are allowed
; nothing here touches a GPU."""
import importlib
import inspect
import json
import os

import pytest

ap = importlib.import_module("mlx_audio_primitives_amd")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_signatures.json")))


def test_every_exported_name_exists():
    missing = [n for n in GOLD["exported"] if not hasattr(ap, n)]
    assert not missing, missing
    assert set(GOLD["exported"]) <= set(ap.__all__)


@pytest.mark.parametrize("name", sorted(GOLD["signatures"]))
def test_signature_matches_reference(name):
    want = GOLD["signatures"][name]
    obj = getattr(ap, name)
    if want.get("class"):
        assert inspect.isclass(obj)
        return
    sig = inspect.signature(obj)
    mine = [p for p in sig.parameters.values() if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
    theirs = want["params"]
    assert [p.name for p in mine[:len(theirs)]] == [p["name"] for p in theirs], name
    for p, q in zip(mine, theirs):
        if q["where"] == "keyword":
            assert p.kind == p.KEYWORD_ONLY, (name, p.name)
        else:
            assert p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD), (name, p.name)
        if q["kind"] == "required":
            assert p.default is inspect.Parameter.empty, (name, p.name)
        elif q["kind"] == "literal":
            assert p.default is not inspect.Parameter.empty, (name, p.name)
            d = list(p.default) if isinstance(p.default, tuple) else p.default
            assert d == q["value"] and type(d) is type(q["value"]), (name, p.name, p.default, q["value"])
    for p in mine[len(theirs):]:                       # our additions must be optional
        assert p.default is not inspect.Parameter.empty, (name, p.name)


@pytest.mark.parametrize("name", sorted(GOLD["ext"]))
def test_ext_entry_point_matches_the_nanobind_module(name):
    """`_ext` mirror (bindings.cpp m.def blocks): same names, argument order and defaults.  Looked up on the
    class: no GPU, no library call."""
    from mlx_audio_primitives_amd._extension import _Ext

    fn = getattr(_Ext, name)
    mine = [p for p in inspect.signature(fn).parameters.values()][1:]          # drop self
    theirs = GOLD["ext"][name]
    assert [p.name for p in mine] == [p["name"] for p in theirs], name
    for p, q in zip(mine, theirs):
        if q["kind"] == "required":
            assert p.default is inspect.Parameter.empty, (name, p.name)
        else:
            assert p.default == q["value"] and type(p.default) is type(q["value"]), (name, p.name, p.default)
