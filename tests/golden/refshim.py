"""Import-time shim that lets the upstream reference (Python >= 3.13 source) be
imported on this container's Python 3.10 -- GOLDEN-GENERATION TOOLING ONLY.

Nothing in the product, the `-m gpu` tests, `smoke()` or `bench.py` imports this
module: `/root/reference` does not exist on the GPU box.  It is used by
`tests/golden/make_golden.py` (run by hand in the build container) to produce
the committed fixtures, and by the optional `test_live_reference.py` tests that
skip themselves when `/root/reference` is absent.

Mechanism: a `sys.meta_path` finder resolves
`happysimulator*` under `/root/reference` and loads the files through a
`SourceFileLoader` whose `get_data()` rewrites PEP-695 generic syntax
(`class N[T](B):`, `def f[T](`) in memory.  No reference source is copied into
this repository and no bytecode is written next to the reference.
"""
from __future__ import annotations

import datetime
import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys
import typing

REFERENCE_ROOT = os.environ.get("HS_REFERENCE_ROOT", "/root/reference")

_CLASS_GENERIC_BASES = re.compile(r"^(\s*)class\s+(\w+)\[([^\]]+)\]\(([^)]*)\):", re.M)
_CLASS_GENERIC_BARE = re.compile(r"^(\s*)class\s+(\w+)\[([^\]]+)\]:", re.M)
_DEF_GENERIC = re.compile(r"^(\s*)def\s+(\w+)\[([^\]]+)\]\(", re.M)


def _param_names(params: str) -> list[str]:
    return [p.split(":")[0].strip() for p in params.split(",") if p.strip()]


def _rewrite(src: str) -> str:
    names: set[str] = set()

    def class_with_bases(m: re.Match) -> str:
        ps = _param_names(m.group(3))
        names.update(ps)
        bases = m.group(4).strip()
        generic = "Generic[" + ", ".join(ps) + "]"
        joined = (bases + ", " + generic) if bases else generic
        return f"{m.group(1)}class {m.group(2)}({joined}):"

    def class_bare(m: re.Match) -> str:
        ps = _param_names(m.group(3))
        names.update(ps)
        return f"{m.group(1)}class {m.group(2)}(Generic[{', '.join(ps)}]):"

    def def_generic(m: re.Match) -> str:
        names.update(_param_names(m.group(3)))
        return f"{m.group(1)}def {m.group(2)}("

    out = _CLASS_GENERIC_BASES.sub(class_with_bases, src)
    out = _CLASS_GENERIC_BARE.sub(class_bare, out)
    out = _DEF_GENERIC.sub(def_generic, out)
    if not names:
        return src
    prelude = "from typing import Generic as Generic, TypeVar as _HsTypeVar\n" + "".join(
        f"{n} = _HsTypeVar({n!r})\n" for n in sorted(names)
    )
    # keep `from __future__` first
    lines = out.split("\n")
    insert_at = 0
    in_doc = False
    for i, line in enumerate(lines):
        s = line.strip()
        if i == 0 and (s.startswith('"""') or s.startswith("'''")):
            q = s[:3]
            if not (len(s) >= 6 and s.endswith(q)):
                in_doc = True
            insert_at = i + 1
            continue
        if in_doc:
            insert_at = i + 1
            if s.endswith('"""') or s.endswith("'''"):
                in_doc = False
            continue
        if s.startswith("from __future__"):
            insert_at = i + 1
        elif s and not s.startswith("#"):
            break
    lines.insert(insert_at, prelude)
    return "\n".join(lines)


class _Loader(importlib.machinery.SourceFileLoader):
    def get_data(self, path):  # noqa: D401
        data = super().get_data(path)
        if path.endswith(".py"):
            return _rewrite(data.decode("utf-8")).encode("utf-8")
        return data


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "happysimulator" and not fullname.startswith("happysimulator."):
            return None
        rel = fullname.replace(".", os.sep)
        base = os.path.join(REFERENCE_ROOT, rel)
        if os.path.isdir(base) and os.path.isfile(os.path.join(base, "__init__.py")):
            file = os.path.join(base, "__init__.py")
            return importlib.util.spec_from_file_location(
                fullname, file, loader=_Loader(fullname, file), submodule_search_locations=[base]
            )
        if os.path.isfile(base + ".py"):
            file = base + ".py"
            return importlib.util.spec_from_file_location(fullname, file, loader=_Loader(fullname, file))
        return None


_installed = False


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "happysimulator", "__init__.py"))


def install() -> None:
    """Make `import happysimulator` resolve to the upstream reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # keep /root/reference clean
    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # type: ignore[attr-defined]
    sys.meta_path.insert(0, _Finder())
    _installed = True
