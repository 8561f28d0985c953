"""Host-side mirror of the reference's ``depth`` package for the GEDepth training hot path
(registry names, constructor kwargs, call protocol and state-dict keys as in qcraftai/gedepth)."""
from ..utils_version import __version__  # noqa: F401
from . import utils  # noqa: F401  (registers SinePositionalEncoding)
