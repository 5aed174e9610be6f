"""One process-wide logger (stderr, timestamped); repeated calls return the same configured object."""
import logging
import sys

_FORMAT = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"
_configured: dict = {}


def setup_logger(name: str = "umbrella_amd", level: int = logging.INFO) -> logging.Logger:
    if name in _configured:
        return _configured[name]
    log = logging.getLogger(name)
    log.setLevel(level)
    log.propagate = False
    if not log.handlers:
        sink = logging.StreamHandler(sys.stderr)
        sink.setFormatter(logging.Formatter(_FORMAT))
        log.addHandler(sink)
    _configured[name] = log
    return log
