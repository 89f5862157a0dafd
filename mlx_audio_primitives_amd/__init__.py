"""Import shim: the package sources live in ``mlx-audio-primitives_amd/`` (the name the
project layout prescribes, which is not a valid Python identifier).  This shim points
the importable name ``mlx_audio_primitives_amd`` at that directory."""

import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "mlx-audio-primitives_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
