"""Import shim: the package directory is named `bert.cpp_amd/` (not an importable identifier), so
`import bert_cpp_amd` extends its search path to that directory."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "bert.cpp_amd"))
