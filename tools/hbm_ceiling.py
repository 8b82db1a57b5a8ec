"""Developer tool: the practical HBM ceiling of this box (msw_core_hbm_stream_rates), several times in a row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msweep_amd.core import Core  # noqa: E402

core = Core(0)
for nb in (1 << 28, 346_000_000, 1 << 30, 1 << 32):
    r, t = core.hbm_stream_rates(nb, 5)
    print(f"{nb / 1e6:9.0f} MB: read-only {r:7.1f} GB/s, triad {t:7.1f} GB/s", flush=True)
core.close()
