"""gym.spaces restatement: only shape/bounds bookkeeping is used by the
reference (babyai/utils/format.py:124-126, babyai/model.py:154)."""
import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Discrete(Space):
    def __init__(self, n):
        self.n = n
        super().__init__((), np.int64)

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        super().__init__(shape, dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    def __repr__(self):
        return "Box" + str(self.shape)


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or {}, **kw)
        super().__init__(None, None)

    def __getitem__(self, key):
        return self.spaces[key]

    def __contains__(self, key):
        return key in self.spaces

    def __repr__(self):
        return "Dict(" + ", ".join("%s:%r" % kv for kv in self.spaces.items()) + ")"
