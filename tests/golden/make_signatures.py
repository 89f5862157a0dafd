"""Writes tests/golden/api_signatures.json: for every public function the reference exports
(mlx_audio_primitives/__init__.py __all__), its parameter names, kinds and literal defaults, taken from the
reference's SOURCE TEXT with `ast` (nothing of the reference is imported or run).  The fixture is data (names
and default literals), the generator stays here.  Run in the build container only:

    python tests/golden/make_signatures.py          # needs /root/reference
"""
import ast
import json
import os
import re

REF = "/root/reference/mlx_audio_primitives"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_signatures.json")


def literal(node):
    if node is None:
        return {"kind": "required"}
    try:
        v = ast.literal_eval(node)
        return {"kind": "literal", "value": v if not isinstance(v, tuple) else list(v)}
    except Exception:
        return {"kind": "expr", "value": ast.unparse(node)}


def function_signatures(path):
    out = {}
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            a = node.args
            pos = a.posonlyargs + a.args
            defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
            params = [{"name": p.arg, "where": "positional", **literal(d)} for p, d in zip(pos, defaults)]
            params += [{"name": p.arg, "where": "keyword", **literal(d)} for p, d in zip(a.kwonlyargs, a.kw_defaults)]
            out[node.name] = {"params": params, "varargs": a.vararg is not None, "varkw": a.kwarg is not None}
        elif isinstance(node, ast.ClassDef):
            out[node.name] = {"class": True}
    return out


def binding_signatures(path):
    """Names, argument names and defaults of the nanobind module `_ext` (csrc/bindings.cpp m.def blocks)."""
    out = {}
    for block in re.split(r"\bm\.def\(", open(path).read())[1:]:
        name = re.search(r'"([a-z_0-9]+)"', block).group(1)
        head = block.split('R"')[0]
        params = []
        for arg, default in re.findall(r'"([A-Za-z_0-9]+)"_a(?:\s*=\s*([^,\n]+))?', head):
            d = default.strip()
            if not d:
                params.append({"name": arg, "kind": "required"})
                continue
            if d == "nb::none()":
                v = None
            elif d in ("true", "false"):
                v = d == "true"
            elif d.startswith('"'):
                v = d.strip('"')
            elif re.fullmatch(r"-?[0-9]+", d):
                v = int(d)
            else:
                v = float(d.rstrip("f"))
            params.append({"name": arg, "kind": "literal", "value": v})
        out[name] = params
    return out


def main():
    init = ast.parse(open(os.path.join(REF, "__init__.py")).read())
    exported = []
    for node in init.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "__all__":
            exported = [e.value for e in node.value.elts]
    found = {}
    for name in sorted(os.listdir(REF)):
        if name.endswith(".py") and name != "__init__.py":
            for fn, sig in function_signatures(os.path.join(REF, name)).items():
                if fn in exported and fn not in found:
                    found[fn] = dict(sig, module=name[:-3])
    missing = [e for e in exported if e not in found]
    ext = binding_signatures("/root/reference/csrc/bindings.cpp")
    json.dump({"exported": exported, "signatures": found, "not_functions": missing, "ext": ext}, open(OUT, "w"),
              indent=1, sort_keys=True)
    print(len(ext), "native entry points")
    print(len(exported), "exported,", len(found), "function signatures,", "others:", missing)


if __name__ == "__main__":
    main()
