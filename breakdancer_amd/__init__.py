"""breakdancer_amd -- MI355X-native anomalous read-pair clustering path of BreakDancerMax.

The product is libbdx.so (hand-written HIP for gfx950 behind the C ABI of include/bdx.h) and the C++
`breakdancer-max` CLI in host/.  This package is the thin Python mirror used by the tests and bench.py."""
from .api import BreakDancer, LibraryConfig, Options, BdxError  # noqa: F401
