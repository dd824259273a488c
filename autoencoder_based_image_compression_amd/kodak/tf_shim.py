"""The sliver of the TensorFlow API that `reconstructing_eae_kodak.py` touches (reference :21, :142, :154, :179, :235):
`tf.Session()` as a context manager with `.run(fetch, feed_dict=...)`, and `tf.reset_default_graph()`.

There is no graph: a "node" is a small object that knows how to evaluate itself on the MI355X from a feed
dictionary. This keeps a driver written against the reference's call surface unchanged in structure.
"""

__version__ = '1.4.0-eae-mi355x-shim'


class Placeholder(object):
    """Stands for `tf.placeholder(tf.float32, shape=...)` (EntropyAutoencoder.py:248-249, IsolatedDecoder.py:101-102)."""

    def __init__(self, shape, name):
        self.shape = tuple(shape)
        self.name = name

    def get_shape(self):
        return self

    def as_list(self):
        return list(self.shape)


class Node(object):
    """A fetchable: `evaluate(feed_dict)` returns a numpy array."""

    def __init__(self, fn, placeholder, name):
        self._fn = fn
        self.placeholder = placeholder
        self.name = name

    def evaluate(self, feed_dict):
        if feed_dict is None or self.placeholder not in feed_dict:
            raise ValueError('You must feed a value for placeholder tensor "{}".'.format(self.placeholder.name))
        value = feed_dict[self.placeholder]
        if tuple(value.shape) != self.placeholder.shape:
            raise ValueError('Cannot feed value of shape {0} for Tensor "{1}", which has shape {2}.'.format(
                tuple(value.shape), self.placeholder.name, self.placeholder.shape))
        return self._fn(value)


class Session(object):
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, (list, tuple)):
            return [node.evaluate(feed_dict) for node in fetches]
        return fetches.evaluate(feed_dict)


def reset_default_graph():
    """The reference tears the TF graph down between the encoder and the decoder (:154, :235). There is no graph here and
    nothing to tear down: a model's buffers go back to torch's caching allocator when its object dies, and the next model
    reuses them (handing them back to the driver with `empty_cache()` only made the next allocations slow)."""
