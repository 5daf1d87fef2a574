from mindspore_rec.train.rec_model import RecModel

__all__ = ["RecModel"]
