"""Logger of the package: same logger name ("onnx_quantize") and ``set_log_level`` entry point as the
reference (_logging.py:29-50), so a host application's logging configuration keeps working."""
from __future__ import annotations

import logging
import sys

__all__ = ["set_log_level"]

_LOGGER_NAME = "onnx_quantize"


def _install_handler() -> None:
    logger = logging.getLogger(_LOGGER_NAME)
    if logger.handlers:
        return
    handler = logging.StreamHandler(sys.stdout)
    handler.setFormatter(logging.Formatter("[%(levelname)s] %(name)s: %(message)s"))
    logger.addHandler(handler)
    logger.setLevel(logging.INFO)
    logger.propagate = False


def set_log_level(level) -> None:
    """Accepts a logging level number or name ("debug", "INFO", ...)."""
    if isinstance(level, str):
        level = getattr(logging, level.upper())
    logging.getLogger(_LOGGER_NAME).setLevel(level)


_install_handler()
