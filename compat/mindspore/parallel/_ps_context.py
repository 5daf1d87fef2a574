"""`mindspore.parallel._ps_context`: the embedding-cache switches HashEmbeddingLookup.__init__ sets
(mindspore_rec/ops/embedding.py:33-38,164-176).  They are recorded; the cache tier itself is
`mindrec_amd.feature_cache.HostBackedTable`."""
_state = {"cache_enable": False, "cache_size": 0, "sparse_format": False, "hash_tables": {}}


def _set_cache_enable(v):
    _state["cache_enable"] = bool(v)


def _cache_enable():
    return _state["cache_enable"]


def _set_cache_size(n):
    _state["cache_size"] = int(n)


def _set_sparse_format(v):
    _state["sparse_format"] = bool(v)


def _insert_hash_table_size(name, cache_vocab_size, embedding_size, vocab_size, param_key=-1):
    _state["hash_tables"][name] = (int(cache_vocab_size), int(embedding_size), int(vocab_size), param_key)


def _is_role_worker():
    from .. import context
    return context.get_ps_context("ms_role") == "MS_WORKER"


def _is_role_pserver():
    from .. import context
    return context.get_ps_context("ms_role") == "MS_PSERVER"


def _is_role_sched():
    from .. import context
    return context.get_ps_context("ms_role") == "MS_SCHED"


def _is_ps_mode():
    from .. import context
    return bool(context.get_ps_context("enable_ps"))
