"""No function of the product package is a re-typed function of the reference.  For every Python file
under thrifty_amd/ that has a same-named file in the reference checkout, every function longer than
five statements is compared with every function of that reference file: both are parsed, identifiers
are normalised (names, arguments and attributes -> one token; docstrings dropped), and the similarity of
the two AST dumps must stay below 0.7.  Runs only where /root/reference exists (the container the
build happens in); the round-5 review found experimental/{carrier,xcorr}_interpolators.py at 0.8-0.98."""
import ast
import difflib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIMIT = 0.7
MIN_STATEMENTS = 5

# (file, function): similarity that is the FORMAT's, not the code's -- each with its reason
UNAVOIDABLE = {
}

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "thrifty")), reason="no reference checkout here")


class _Normalise(ast.NodeTransformer):
    def visit_Name(self, node):
        return ast.copy_location(ast.Name(id="v", ctx=node.ctx), node)

    def visit_arg(self, node):
        return ast.copy_location(ast.arg(arg="v", annotation=None), node)

    def visit_Attribute(self, node):
        self.generic_visit(node)
        node.attr = "a"
        return node

    def visit_FunctionDef(self, node):
        self.generic_visit(node)
        node.name = "f"
        node.decorator_list = []
        if node.body and isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], "value", None), ast.Constant) \
                and isinstance(node.body[0].value.value, str):
            node.body = node.body[1:] or [ast.Pass()]
        return node

    def visit_keyword(self, node):
        self.generic_visit(node)
        node.arg = "k"
        return node


def _parse(path):
    src = open(path, encoding="utf-8", errors="replace").read()
    try:
        return ast.parse(src)
    except SyntaxError:
        # Python 2 sources: the few constructs that stop the Python 3 parser, textually
        import re
        src = re.sub(r"^(\s*)print (?!\()(.*)$", r"\1print(\2)", src, flags=re.M)
        src = re.sub(r"except (\w+), (\w+):", r"except \1 as \2:", src)
        try:
            return ast.parse(src)
        except SyntaxError:
            return None


def _statements(fn):
    return sum(isinstance(n, ast.stmt) for n in ast.walk(fn)) - 1


def _functions(tree):
    return [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]


def _dump(fn):
    import copy
    return ast.dump(_Normalise().visit(copy.deepcopy(fn)), annotate_fields=False)


# product files whose reference counterpart has another name (code that was split out of a mirror)
ALSO = {"detect_cli.py": ["detect.py"], "stages.py": ["carrier_sync.py", "soa_estimator.py", "detect.py"],
        "fastdet.py": ["detect.py"], "parallel.py": ["detect.py"], "synth.py": ["gold.py", "template_generate.py"]}


def _reference_files():
    index = {}
    for dirpath, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".py"):
                index.setdefault(f, []).append(os.path.join(dirpath, f))
    for ours, theirs in ALSO.items():
        for name in theirs:
            index.setdefault(ours, []).extend(index.get(name, []))
    return index


def similarities():
    index = _reference_files()
    out = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "thrifty_amd")):
        for f in files:
            if not f.endswith(".py") or f not in index or f == "__init__.py":
                continue
            ours = _parse(os.path.join(dirpath, f))
            mine = [(fn, _dump(fn)) for fn in _functions(ours) if _statements(fn) > MIN_STATEMENTS]
            for ref_path in index[f]:
                theirs = _parse(ref_path)
                if theirs is None:
                    continue
                ref_dumps = [(fn.name, _dump(fn)) for fn in _functions(theirs)]
                for fn, d in mine:
                    for name, rd in ref_dumps:
                        r = difflib.SequenceMatcher(None, d, rd, autojunk=False).ratio()
                        out.append((r, os.path.relpath(os.path.join(dirpath, f), ROOT), fn.name,
                                    os.path.relpath(ref_path, REF), name))
    return sorted(out, reverse=True)


def test_no_function_is_a_normalised_copy_of_a_reference_function():
    sims = similarities()
    assert sims, "nothing compared: no same-named files?"
    bad = [s for s in sims if s[0] >= LIMIT and (s[1], s[2]) not in UNAVOIDABLE]
    assert not bad, "\n".join("%.2f  %s:%s  ~  %s:%s" % s for s in bad[:20])


if __name__ == "__main__":
    for s in similarities()[:25]:
        print("%.2f  %s:%s  ~  %s:%s" % s)
