"""Import shim: the package directory is named `gf-orb-slam2_amd/` (not a Python identifier);
this module makes it importable as `gf_orb_slam2_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gf-orb-slam2_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
