"""Import shim: the product package directory is `zksnark-finalproject_amd/` (not a valid Python identifier);
`import zksnark_finalproject_amd` resolves its submodules from there."""
import os

__path__.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zksnark-finalproject_amd"))
from ._lib import LIB_PATH, SIGNATURES, Zkg16Error, load  # noqa: E402,F401
from .device import Device  # noqa: E402,F401
