"""Import alias: the product package lives in `cu-sdr-collection_amd/` (the hyphen is part of
the upstream repository name and is not a legal Python identifier).  This shim makes it
importable as `cu_sdr_collection_amd` by pointing the package search path at that directory
and executing its `__init__.py` in this module's namespace."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "cu-sdr-collection_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
