"""`mindspore.parallel`: the private helpers mindspore_rec imports."""
