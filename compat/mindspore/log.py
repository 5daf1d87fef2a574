"""`mindspore.log` (imported as `logger`, mindspore_rec/train/rec_model.py:24)."""
import logging

_l = logging.getLogger("mindspore")
debug, info, warning, error, critical = _l.debug, _l.info, _l.warning, _l.error, _l.critical
