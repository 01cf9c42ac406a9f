"""HIP (gfx950) back end: ctypes binding (`_lib`) and torch-tensor level operators (`ops`)."""
from ._lib import OqHipError, OqHipMissing, load  # noqa: F401
