"""Logging setup: stdout + <output dir>/boolsi.log, level-dependent line format (reference log.py:8-67)."""
import logging
import os
import sys

from .constants import log_date_format


class _LevelFormatter(logging.Formatter):
    def __init__(self):
        super().__init__()
        self._plain = logging.Formatter('%(asctime)s %(message)s', datefmt=log_date_format)
        self._tagged = logging.Formatter('%(asctime)s %(levelname)s %(message)s', datefmt=log_date_format)

    def format(self, record):
        tagged = record.levelno in (logging.WARNING, logging.ERROR)
        return (self._tagged if tagged else self._plain).format(record)


def configure_logging(output_directory):
    os.makedirs(output_directory, exist_ok=True)
    root = logging.getLogger()
    for handler in list(root.handlers):
        root.removeHandler(handler)
    root.setLevel(logging.INFO)
    for handler in (logging.StreamHandler(sys.stdout),
                    logging.FileHandler(os.path.join(output_directory, 'boolsi.log'), mode='w')):
        handler.setLevel(logging.INFO)
        handler.setFormatter(_LevelFormatter())
        root.addHandler(handler)
