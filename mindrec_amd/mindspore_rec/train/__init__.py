from .rec_model import RecModel

__all__ = ["RecModel"]
