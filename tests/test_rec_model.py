"""RecModel.online_train contracts.  The first three tests restate the reference's own CI tests
(ci/st/online_learning/test_online_learning.py:54-114) against this implementation: same call,
same exception type, same message substring."""
import numpy as np
import pytest

from mindrec_amd import context, nn
from mindrec_amd.mindspore_rec import RecModel
from mindrec_amd.mindspore_rec.train.callback import Callback, TimeMonitor


class StreamingDataset:
    """Unbounded stream like the reference's fixture (test_online_learning.py:43-51), batched by 100."""

    def __init__(self, limit=None):
        self.limit = limit
        self.resets = 0

    def __iter__(self):
        i = 0
        while self.limit is None or i < self.limit:
            i += 1
            yield (np.ones((100, 39), np.int32),)

    def get_dataset_size(self):
        return 2**20 - 1

    def reset(self):
        self.resets += 1


class Net:
    training = True

    def __init__(self):
        self.calls = 0

    def train(self, mode=True):
        self.training = mode

    def __call__(self, ids):
        self.calls += 1
        return float(ids.sum())


@pytest.fixture(autouse=True)
def _gpu_target():
    context.set_context(mode=context.GRAPH_MODE, device_target="GPU")
    yield
    context.set_context(device_target="GPU")


def test_online_learning_api_sink_size_is_negative():
    model = RecModel(Net())
    with pytest.raises(ValueError) as exc_info:
        model.online_train(StreamingDataset(), dataset_sink_mode=True, sink_size=-1)
    assert "The input value must be int and must > 0" in str(exc_info.value)


def test_online_learning_api_sink_size_not_equal_one():
    model = RecModel(Net())
    with pytest.raises(ValueError) as exc_info:
        model.online_train(StreamingDataset(), dataset_sink_mode=True, sink_size=100)
    assert "The sink_size parameter only support value of 1" in str(exc_info.value)


def test_online_learning_api_data_sink_mode_not_bool():
    model = RecModel(Net())
    with pytest.raises(TypeError) as exc_info:
        model.online_train(StreamingDataset(), dataset_sink_mode="valid")
    assert "The input value must be a bool, but got str" in str(exc_info.value)


def test_graph_cell_refused_in_sink_mode():
    class G(nn.GraphCell):
        def construct(self, x):
            return x
    with pytest.raises(ValueError, match="not supported when training with a GraphCell"):
        RecModel(G()).online_train(StreamingDataset(), dataset_sink_mode=True)


class Recorder(Callback):
    def __init__(self, stop_after):
        self.stop_after, self.events, self.params = stop_after, [], None

    def begin(self, rc): self.events.append("begin")
    def epoch_begin(self, rc): self.events.append("epoch_begin")
    def step_begin(self, rc): self.events.append("step_begin")

    def step_end(self, rc):
        p = rc.original_args()
        self.events.append(("step_end", p.cur_epoch_num, p.cur_step_num))
        self.params = dict(p)
        if p.cur_step_num >= self.stop_after:
            rc.request_stop()

    def epoch_end(self, rc): self.events.append("epoch_end")
    def end(self, rc): self.events.append("end")


@pytest.mark.parametrize("sink", [True, False])
def test_online_train_loop_and_callbacks(sink):
    net, ds, rec = Net(), StreamingDataset(limit=3), Recorder(stop_after=7)
    tm = TimeMonitor()
    RecModel(net).online_train(ds, callbacks=[rec, tm], dataset_sink_mode=sink, sink_size=1)
    assert net.calls == 7 and len(tm.step_ms) == 7
    steps = [e for e in rec.events if isinstance(e, tuple)]
    # 3 batches per epoch, unbounded epochs: step counter runs on, epoch counter advances (rec_model.py:211-249)
    assert steps == [("step_end", 1, 1), ("step_end", 1, 2), ("step_end", 1, 3), ("step_end", 2, 4), ("step_end", 2, 5),
                     ("step_end", 2, 6), ("step_end", 3, 7)]
    assert rec.events[0] == "begin" and rec.events[-1] == "end" and rec.events.count("epoch_begin") == 3
    assert rec.params["dataset_sink_mode"] is sink
    assert rec.params["batch_num"] == (1 if sink else 2**20 - 1)          # :166-171
    assert rec.params["net_outputs"] == 3900.0 and rec.params["train_network"] is net
    if not sink:
        assert ds.resets == 3                                              # train_dataset.reset() per epoch (:245)


def test_cpu_target_forces_feed_mode():
    context.set_context(device_target="CPU")
    rec = Recorder(stop_after=2)
    RecModel(Net()).online_train(StreamingDataset(), callbacks=rec, dataset_sink_mode=True, sink_size=1)
    assert rec.params["dataset_sink_mode"] is False                        # rec_model.py:179-186


def test_callbacks_must_be_callback_objects():
    with pytest.raises(TypeError):
        RecModel(Net()).online_train(StreamingDataset(limit=1), callbacks=[object()], dataset_sink_mode=False)


def test_hash_embedding_lookup_cache_without_ps_raises():
    """mindspore_rec/ops/embedding.py:105-110: vocab_cache_size > 0 needs parameter-server mode."""
    from mindrec_amd.mindspore_rec import HashEmbeddingLookup
    context.reset_ps_context()
    with pytest.raises(RuntimeError, match="only support in parameter server training mode"):
        HashEmbeddingLookup(16, vocab_cache_size=1000)
    with pytest.raises(TypeError):
        HashEmbeddingLookup(16, sparse="yes")
    with pytest.raises(ValueError):
        HashEmbeddingLookup(16, vocab_cache_size=-1)
