#!/usr/bin/env python3
"""Which attributes of its detection / preprocessing / stats modules the reference's callers use: data for
tests/test_integration_recipe.py (authoring container only; /root/reference does not travel).

    python tests/golden/make_call_sites.py

Parses chromosight/cli/chromosight.py and chromosight/utils/contacts_map.py with `ast`: `import ... as <alias>` of the
three modules, then every `<alias>.<attr>` and every `from <module> import <name>`; writes reference_call_sites.json
{module: {attr: [file:line, ...]}}."""
import ast
import json
import pathlib

REF = pathlib.Path("/root/reference/chromosight")
FILES = [REF / "cli" / "chromosight.py", REF / "utils" / "contacts_map.py"]
MODULES = {"chromosight.utils.detection": "detection", "chromosight.utils.preprocessing": "preprocessing",
           "chromosight.utils.stats": "stats"}


def main():
    out = {m: {} for m in MODULES.values()}
    for path in FILES:
        tree = ast.parse(path.read_text())
        alias = {}
        rel = f"{path.parent.name}/{path.name}"
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                for a in node.names:
                    if a.name in MODULES:
                        alias[a.asname or a.name] = MODULES[a.name]
            elif isinstance(node, ast.ImportFrom) and node.module in MODULES:
                for a in node.names:
                    out[MODULES[node.module]].setdefault(a.name, []).append(f"{rel}:{node.lineno}")
            elif isinstance(node, ast.ImportFrom) and node.level == 1 and node.module is None and path.parent.name == "utils":
                for a in node.names:                     # from . import preprocessing as preproc
                    full = f"chromosight.utils.{a.name}"
                    if full in MODULES:
                        alias[a.asname or a.name] = MODULES[full]
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in alias:
                out[alias[node.value.id]].setdefault(node.attr, []).append(f"{rel}:{node.lineno}")
    for m in out:
        out[m] = {k: sorted(set(v)) for k, v in sorted(out[m].items())}
    dest = pathlib.Path(__file__).with_name("reference_call_sites.json")
    dest.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
