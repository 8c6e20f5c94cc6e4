"""Small shared pieces: constants, Mode, logger (reference: neural_compressor/common/utils/{constants,logger}.py)."""

import logging
import os
from enum import Enum

# algorithm names (reference common/utils/constants.py:28-33)
RTN, GPTQ, AWQ = "rtn", "gptq", "awq"
SMOOTH_QUANT = "smooth_quant"  # constants.py:31
DEFAULT_WHITE_LIST = "*"  # constants.py:22
EMPTY_WHITE_LIST = None  # constants.py:23


class Mode(Enum):
    """Two-phase protocol of every algorithm entry (reference common/utils/constants.py:55)."""

    PREPARE = "prepare"
    CONVERT = "convert"
    QUANTIZE = "quantize"
    LOAD = "load"


def _make_logger():
    log = logging.getLogger("neural_compressor_amd")
    if not log.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s [%(levelname)s][%(filename)s:%(lineno)d] %(message)s", "%Y-%m-%d %H:%M:%S"))
        log.addHandler(h)
        log.propagate = False
    log.setLevel(os.environ.get("LOGLEVEL", "WARNING").upper())  # same env switch as the reference logger.py:64
    return log


logger = _make_logger()
