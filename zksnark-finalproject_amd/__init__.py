"""zkg16 — MI355X-native Groth16 (BLS12-381) prove hot path behind the ark-groth16 prover surface used by
ArielElb/zkSnark-FinalProject (src/arkworks/backend/matrix_proof.rs:139-140).  See DESIGN.md."""
from ._lib import LIB_PATH, SIGNATURES, Zkg16Error, load  # noqa: F401
from .device import Device  # noqa: F401
