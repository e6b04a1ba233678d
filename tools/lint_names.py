"""Poor man's pyflakes (none is installed): report names that are loaded in a module but never bound in it, imported, or a
builtin.  usage: python tools/lint_names.py <files...>   (exit 1 on findings) — run before spending a GPU call."""
import ast
import builtins
import sys


def check(path):
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(node.name)
            if not isinstance(node, ast.ClassDef):
                for a in node.args.args + node.args.kwonlyargs + node.args.posonlyargs:
                    bound.add(a.arg)
                for a in (node.args.vararg, node.args.kwarg):
                    if a:
                        bound.add(a.arg)
        elif isinstance(node, ast.Lambda):
            for a in node.args.args + node.args.kwonlyargs:
                bound.add(a.arg)
            for a in (node.args.vararg, node.args.kwarg):
                if a:
                    bound.add(a.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            bound.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
        elif isinstance(node, (ast.Global, ast.Nonlocal)):
            bound.update(node.names)
    bad = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
            bad.append(f"{path}:{node.lineno}: undefined name {node.id}")
    return bad


if __name__ == "__main__":
    out = [m for f in sys.argv[1:] for m in check(f)]
    print("\n".join(out) if out else "names ok")
    sys.exit(1 if out else 0)
