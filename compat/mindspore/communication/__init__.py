"""`mindspore.communication`."""
from . import management  # noqa: F401
from .management import GlobalComm, get_group_size, get_local_rank, get_rank, init, release  # noqa: F401
