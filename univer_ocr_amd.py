"""`import univer_ocr_amd` -> the package directory `univer-ocr_amd/` (the name the project layout prescribes is
not a valid Python identifier).  A module that sets `__path__` IS a package to the import system, so
`import univer_ocr_amd.nn.layers` resolves inside that directory; nothing is executed or copied here."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'univer-ocr_amd')]
__version__ = '0.2.0'
