"""Import shim: makes the package directory `univer-ocr_amd/` (not a valid Python identifier)
importable as `univer_ocr_amd` -- `import univer_ocr_amd.nn.layers`, etc."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'univer-ocr_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f
