"""hifir_amd -- MI355X-native preconditioner-apply path for HIFIR multilevel ILU hierarchies.

Host-side mirror (Python over the C ABI of include/hifir_amd.h) of the reference's operator
interface for this path: `HIF` follows hif::HIF<> (src/hif/builder.hpp:109) -- solve, hifir,
levels, nnz, rank, schur_size -- with the factored hierarchy resident in HBM.
"""
from ._lib import build, lib  # noqa: F401
from .hif import HIF, HifAmdError  # noqa: F401

__version__ = "0.1.0"
