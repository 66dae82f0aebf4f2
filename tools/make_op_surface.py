"""Snapshot of the reference's op surface for the hot path -> tests/golden/reference_op_surface.json.
Runs only where /root/reference exists.  Three sources, parsed as text / AST (nothing is imported):
  * csrc/torch_bindings.cpp: every `X.def("schema string")` of the namespaces _C, _C_cache_ops, _C_cuda_utils,
    _C_custom_ar, and every `X.def("name", &fn)` whose schema torch infers from the C++ signature;
  * csrc/ops.h, csrc/cache.h, csrc/cuda_utils.h, csrc/custom_all_reduce declarations in ops.h: the C++ signatures
    of the inferred ones, turned into schema strings the way torch's inference does (Tensor / int / float / bool /
    str, `Tensor&` out-parameters are not annotated by inference);
  * vllm/_custom_ops.py: function names and parameter names (ast).
tests/test_op_surface.py compares this snapshot with neural_magic_vllm_amd's registrations; when the reference
is present it also re-derives the snapshot and requires it to be current."""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "reference_op_surface.json")

NAMESPACES = {"ops": "_C", "cache_ops": "_C_cache_ops", "cuda_utils": "_C_cuda_utils", "custom_ar": "_C_custom_ar"}
CTYPES = [(r"const\s+torch::Tensor\s*&", "Tensor"), (r"torch::Tensor\s*&", "Tensor"), (r"torch::Tensor", "Tensor"),
          (r"const\s+c10::optional<torch::Tensor>\s*&", "Tensor?"), (r"c10::optional<torch::Tensor>", "Tensor?"),
          (r"const\s+std::optional<torch::Tensor>\s*&", "Tensor?"),
          (r"std::pair<std::vector<uint8_t>,\s*std::vector<int64_t>>", "(int[], int[])"),
          (r"const\s+std::vector<std::vector<int64_t>>\s*&", "int[][]"),
          (r"std::tuple<torch::Tensor,\s*std::vector<int64_t>>", "(Tensor, int[])"),
          (r"const\s+std::vector<torch::Tensor>\s*&", "Tensor[]"), (r"std::vector<torch::Tensor>\s*&?", "Tensor[]"),
          (r"const\s+std::vector<std::string>\s*&", "str[]"), (r"const\s+std::vector<int64_t>\s*&", "int[]"),
          (r"const\s+std::string\s*&", "str"), (r"std::string", "str"), (r"int64_t", "int"), (r"\bint\b", "int"),
          (r"\bdouble\b", "float"), (r"\bfloat\b", "float"), (r"\bbool\b", "bool"), (r"fptr_t", "int")]


def norm(s: str) -> str:
    return re.sub(r"\s+", " ", s).replace("( ", "(").replace(" )", ")").strip()


def cpp_decls():
    """name -> (return type, [(type, name)]) for every function declared in the reference's op headers"""
    text = ""
    for h in ("csrc/ops.h", "csrc/cache.h", "csrc/cuda_utils.h"):
        text += re.sub(r"//[^\n]*", "", open(os.path.join(REF, h)).read()) + "\n"
    text = re.sub(r"#[^\n]*", "", text)
    decls = {}
    for m in re.finditer(r"([\w:<>,\s&]+?)\s+(\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = norm(m.group(1)), m.group(2), norm(m.group(3))
        args = []
        ok = True
        for prm in [p for p in params.split(",") if p.strip()] if params not in ("", "void") else []:
            prm = prm.strip()
            mm = re.match(r"(.+?)\s*(\w+)$", prm)
            ty, an = (mm.group(1).strip(), mm.group(2)) if mm else (prm, "")
            for pat, rep in CTYPES:
                if re.fullmatch(pat, ty):
                    ty = rep
                    break
            else:
                ok = False
            args.append((ty, an))
        for pat, rep in CTYPES + [(r"void", "()")]:
            if re.fullmatch(pat, ret):
                ret = rep
                break
        if ok:
            decls[name] = (ret, args)
    return decls


def bindings():
    src = open(os.path.join(REF, "csrc/torch_bindings.cpp")).read()
    src = re.sub(r"//[^\n]*", "", src)
    decls = cpp_decls()
    out = {ns: {} for ns in NAMESPACES.values()}
    # X.def("....");  or  X.def("name", &fn);
    for m in re.finditer(r"\b(\w+)\.def\(\s*((?:\"[^\"]*\"\s*)+)(?:,\s*&(\w+))?\s*\)\s*;", src, flags=re.S):
        var, lit, fn = m.group(1), m.group(2), m.group(3)
        if var not in NAMESPACES:
            continue
        s = norm("".join(re.findall(r"\"([^\"]*)\"", lit)))
        if fn is None:
            name = s.split("(")[0]
            out[NAMESPACES[var]][name] = dict(schema=s, inferred=False)
        else:
            if fn not in decls:      # an op outside the hot path whose C++ types this parser does not map
                continue
            ret, args = decls[fn]
            schema = f"{s}({', '.join(f'{t} {n}' for t, n in args)}) -> {ret}"
            out[NAMESPACES[var]][s] = dict(schema=schema, inferred=True)
    return out


def custom_ops():
    tree = ast.parse(open(os.path.join(REF, "vllm/_custom_ops.py")).read())
    fns = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            a = node.args
            fns[node.name] = [x.arg for x in a.posonlyargs + a.args] + [x.arg for x in a.kwonlyargs]
    return fns


def snapshot():
    return dict(torch_bindings=bindings(), custom_ops=custom_ops())


if __name__ == "__main__":
    snap = snapshot()
    with open(OUT, "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)
    n = sum(len(v) for v in snap["torch_bindings"].values())
    print(f"{OUT}: {n} ops in {list(snap['torch_bindings'])}, {len(snap['custom_ops'])} _custom_ops functions")
