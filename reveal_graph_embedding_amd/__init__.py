"""Import shim: the package sources live in ``reveal-graph-embedding_amd/`` (the
directory name the build contract fixes; a hyphen is not importable), this makes
them importable as ``reveal_graph_embedding_amd``."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "reveal-graph-embedding_amd")
__path__.insert(0, _src)
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _os, _f, _src
