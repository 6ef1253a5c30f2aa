"""In-memory loader for the upstream reference (dev container only).

TEST INFRASTRUCTURE - never imported by the product path, never shipped to the
GPU box.  It reads /root/reference/src/*.py at run time, rewrites the Python 2
constructs in memory (print statements, integer '/'), stubs the modules that are
not installable here (tensorflow, cv2) and exec()s the result.  No reference
text is stored in this repository; only the numeric outputs of running it are
committed (tests/golden/*.npz, made by gen_golden.py).

Recipe:
  1. lib2to3 fix_print         (Py2 print statement -> function)
  2. ast rewrite of every '/'  -> _py2div(l, r): floor-div when both operands
     are (numpy) integers, true division otherwise  == Python 2 semantics
  3. stub tensorflow / cv2 / tqdm / model in sys.modules
"""
import ast
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np

REF_SRC = os.environ.get("MCCNN_REFERENCE_SRC", "/root/reference/src")


def _py2div(l, r):
    if isinstance(l, (int, np.integer)) and isinstance(r, (int, np.integer)) \
            and not isinstance(l, bool) and not isinstance(r, bool):
        return l // r
    return l / r


class _DivRewriter(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()),
                         args=[node.left, node.right], keywords=[]), node)
        return node


def _load(name, extra_modules):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    path = os.path.join(REF_SRC, name + ".py")
    with open(path) as f:
        src = f.read()
    src3 = str(tool.refactor_string(src, name))
    tree = _DivRewriter().visit(ast.parse(src3, filename=path))
    ast.fix_missing_locations(tree)
    mod = types.ModuleType("ref_" + name)
    mod.__dict__["_py2div"] = _py2div
    saved = {k: sys.modules.get(k) for k in extra_modules}
    sys.modules.update(extra_modules)
    try:
        exec(compile(tree, path, "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def available():
    return os.path.isfile(os.path.join(REF_SRC, "process_functional.py"))


def load_reference():
    """Returns (process_functional, util) modules of the reference, py3-shimmed."""
    stub_tf = types.ModuleType("tensorflow")
    stub_cv2 = types.ModuleType("cv2")
    stub_tqdm = types.ModuleType("tqdm")
    stub_tqdm.tqdm = lambda x, *a, **k: x
    stub_model = types.ModuleType("model")
    stub_model.NET = None
    util = _load("util", {"cv2": stub_cv2})
    pf = _load("process_functional", {
        "tensorflow": stub_tf, "cv2": stub_cv2, "tqdm": stub_tqdm,
        "model": stub_model, "util": util})
    return pf, util


@contextlib.contextmanager
def quiet():
    """The reference prints a line per disparity / iteration; silence it."""
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yield
    finally:
        sys.stdout = old
