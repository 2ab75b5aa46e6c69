"""Import alias: ``progressive_stable_diffusion_amd`` -> ``progressive-stable-diffusion_amd/``.

The product directory carries the reference repository's hyphenated name, which is not a
Python identifier; this package only redirects ``__path__`` there.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "progressive-stable-diffusion_amd")]
exec(compile(open(_os.path.join(__path__[0], "__init__.py")).read(),
             _os.path.join(__path__[0], "__init__.py"), "exec"))
