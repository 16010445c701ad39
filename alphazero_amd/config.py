"""YAML config loader with the reference's layout (alphazero/config.py:4-33,
alphazero/config/*.yaml): sections game / model / loss / training / selfplay, classes
named by module + name + kwargs.  Module paths written for the reference
(`alphazero.games.connectfour`, ...) resolve to this package's equivalents."""
import importlib

import yaml

_ALIASES = {
    'alphazero.games.connectfour': 'alphazero_amd.games.connectfour',
    'alphazero.games.tictactoe': 'alphazero_amd.games.tictactoe',
    'alphazero.models': 'alphazero_amd.models',
    'alphazero.loss': 'alphazero_amd.loss',
}
_CLASS_HOME = {'AlphaZeroC4': 'alphazero_amd.models'}


def _resolve(module, name):
    module = _CLASS_HOME.get(name, _ALIASES.get(module, module))
    return getattr(importlib.import_module(module), name)


def _instance(spec, overrides=None):
    kwargs = dict(spec.get('kwargs', {}))
    kwargs.update(overrides or {})
    return _resolve(spec['module'], spec['name'])(**kwargs)


class AlphaZeroConfig:
    def __init__(self, path=None, conf=None):
        if conf is None:
            with open(path) as f:
                conf = yaml.safe_load(f)
        self._config_path = path
        self._conf = conf

    def Game(self, **kwargs):
        return _instance(self._conf['game'], kwargs)

    def Model(self):
        return _instance(self._conf['model'])

    def Loss(self):
        return _instance(self._conf['loss'])

    @property
    def game_class(self):
        s = self._conf['game']
        return _resolve(s['module'], s['name'])

    def __getattr__(self, attr):
        assert attr in ('selfplay', 'training'), attr
        return self._conf[attr]
