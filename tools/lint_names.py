#!/usr/bin/env python3
"""Undefined-name check for the package (there is no GPU in the build container, so a NameError on a HIP-only path would
otherwise surface on the GPU box): every name a function reads without binding it must be a module global or a builtin.
    python tools/lint_names.py [files...]        exit code 1 if anything is unresolved"""
import builtins
import os
import symtable
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = set(top.get_identifiers()) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []

    def walk(tab, enclosing):
        bound = {s.get_name() for s in tab.get_symbols() if s.is_assigned() or s.is_parameter() or s.is_imported() or s.is_namespace()}
        for s in tab.get_symbols():
            n = s.get_name()
            if s.is_referenced() and not (s.is_assigned() or s.is_parameter() or s.is_imported() or s.is_namespace()):
                if s.is_free() or n in enclosing or n in module_names:
                    continue
                bad.append((tab.get_name(), tab.get_lineno(), n))
        for c in tab.get_children():
            walk(c, enclosing | bound if tab.get_type() == "function" else enclosing)
    for c in top.get_children():
        walk(c, set())
    return bad


def main():
    pkg = os.path.join(REPO, "gnnome_assembly_amd")
    files = sys.argv[1:] or ([os.path.join(pkg, f) for f in sorted(os.listdir(pkg)) if f.endswith(".py")]
                             + [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")]
                             + [os.path.join(REPO, d, f) for d in ("tests", "tools") for f in sorted(os.listdir(os.path.join(REPO, d)))
                                if f.endswith(".py")])
    rc = 0
    for f in files:
        for fn, line, name in check(f):
            print(f"{os.path.relpath(f, REPO)}:{line}: in {fn}(): name {name!r} is not defined")
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
