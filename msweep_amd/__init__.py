"""msweep_amd -- MI355X-native abundance-estimation core for mSWEEP's hot path
(likelihood build, RCG / EM optimiser, bootstrap).  See DESIGN.md and INTEGRATION.md."""
from .core import ALGO_EM, ALGO_RCG, PREC_DOUBLE, PREC_FLOAT, Core, MswError, load_library  # noqa: F401

__version__ = "0.1.0"
