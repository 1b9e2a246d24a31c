"""gym.envs.registration restatement: id -> 'module:Class' entry points
(reference: babyai/levels/levelgen.py:480-486)."""
import importlib
from .. import error


class EnvSpec(object):
    def __init__(self, id, entry_point=None, kwargs=None, **_ignored):
        self.id = id
        self.entry_point = entry_point
        self._kwargs = kwargs or {}

    def make(self, **kwargs):
        kw = dict(self._kwargs)
        kw.update(kwargs)
        if callable(self.entry_point):
            env = self.entry_point(**kw)
        else:
            mod_name, attr = self.entry_point.split(':')
            cls = getattr(importlib.import_module(mod_name), attr)
            env = cls(**kw)
        env.unwrapped.spec = self
        return env


class EnvRegistry(object):
    def __init__(self):
        self.env_specs = {}

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise error.Error('Cannot re-register id: {}'.format(id))
        self.env_specs[id] = EnvSpec(id, **kwargs)

    def make(self, id, **kwargs):
        if id not in self.env_specs:
            raise error.UnregisteredEnv('No registered env with id: {}'.format(id))
        return self.env_specs[id].make(**kwargs)

    def all(self):
        return self.env_specs.values()


registry = EnvRegistry()


def register(id, **kwargs):
    return registry.register(id, **kwargs)


def make(id, **kwargs):
    return registry.make(id, **kwargs)
