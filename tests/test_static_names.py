"""Static check of the host package: every name a function of cvvae_amd/*.py loads is one of its arguments / locals, a module-level
name or a builtin.  Most of the package only executes on a GPU box, so an undefined name in a branch the CPU suite never runs
(e.g. a variable of one engine function pasted into its sibling) would otherwise surface at round end."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stored(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n is not node:
            out.add(n.name)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            out.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


def test_no_undefined_names_in_host_package():
    bad = []
    for f in sorted(glob.glob(os.path.join(ROOT, "cvvae_amd", "*.py"))) + [os.path.join(ROOT, "bench.py"),
                                                                         os.path.join(ROOT, "__graft_entry__.py")]:
        tree = ast.parse(open(f).read())
        top = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for n in tree.body:
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                top.add(n.name)
            else:
                top |= _stored(n)
        funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
        for c in (n for n in tree.body if isinstance(n, ast.ClassDef)):
            cls_names = _stored(c)  # class attributes are reachable unqualified only inside the class body, methods use self.
            funcs += [m for m in c.body if isinstance(m, ast.FunctionDef)]
        for fn in funcs:
            local = _stored(fn)
            for n in ast.walk(fn):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local and n.id not in top:
                    bad.append(f"{os.path.relpath(f, ROOT)}:{n.lineno}: undefined name '{n.id}' in {fn.name}()")
    assert not bad, "\n".join(bad)
