"""Logging seam: inside LichtFeld Studio the plugin logs through ``lichtfeld.log`` (upstream
core/pipeline.py:10,156,...); outside the host (tests, bench, CLI on a headless box) the same calls go
to the standard ``logging`` module."""
from __future__ import annotations

import logging


class _StdLog:
    def __init__(self) -> None:
        self._log = logging.getLogger("lfd_densify")

    def info(self, msg) -> None:
        self._log.info(str(msg))

    def warn(self, msg) -> None:
        self._log.warning(str(msg))

    def error(self, msg) -> None:
        self._log.error(str(msg))

    def debug(self, msg) -> None:
        self._log.debug(str(msg))


def get_log():
    try:
        import lichtfeld as lf  # provided by the host application
        return lf.log
    except Exception:
        return _StdLog()


log = get_log()
