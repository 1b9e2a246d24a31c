"""Stub for `blosc` (reference: babyai/utils/demos.py:5,53) -- demo files are
out of scope; only importability of `babyai` matters.  TEST INFRASTRUCTURE."""
import pickle


def pack_array(arr):
    return pickle.dumps(arr)


def unpack_array(data):
    return pickle.loads(data)
