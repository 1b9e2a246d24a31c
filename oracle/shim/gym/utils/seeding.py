"""gym.utils.seeding restatement (gym <= 0.21, the `env.seed(s)` era that the
reference's call sites require: scripts/train_rl.py:59, babyai/evaluate.py:105-106).

np_random(seed): seed -> sha512(str(seed)) -> first 8 bytes as a little-endian
big int -> list of uint32 words (low word first, leading zeros dropped) ->
numpy.random.RandomState().seed(words)  (MT19937 init_by_array).
"""
import hashlib
import os
import struct

import numpy as np

from .. import error


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise error.Error('Seed must be a non-negative integer or omitted, not {}'.format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    digest = hashlib.sha512(str(seed).encode('utf8')).digest()
    return _bigint_from_bytes(digest[:max_bytes])


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, str):
        a = a.encode('utf8')
        a += hashlib.sha512(a).digest()
        a = _bigint_from_bytes(a[:max_bytes])
    elif isinstance(a, (int, np.integer)):
        a = int(a) % 2 ** (8 * max_bytes)
    else:
        raise error.Error('Invalid type for seed: {} ({})'.format(type(a), a))
    return a


def _bigint_from_bytes(data):
    sizeof_int = 4
    padding = sizeof_int - len(data) % sizeof_int
    data += b'\0' * padding
    int_count = len(data) // sizeof_int
    unpacked = struct.unpack("{}I".format(int_count), data)
    accum = 0
    for i, val in enumerate(unpacked):
        accum += 2 ** (sizeof_int * 8 * i) * val
    return accum


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise error.Error('Seed must be non-negative, not {}'.format(bigint))
    elif bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints
