"""ModelServer with the reference's interface (alphazero/model_server.py:7-48) and no Redis.

Inside one process the latest weights are simply kept (update / latest / reset); between the
processes of a multi-GPU job `distributed.broadcast_model` moves them (one RCCL broadcast), so the
trainer rank calls `update` and every rank calls `sync` before `latest`.  Checkpoints are written in
the reference's format -- `{'model_version', 'model_state_dict'}` with the aliased `_tower.i.*` keys,
`<logdir>/ckpt/<version:06d>.pt` -- and `load_ckpt` reads either side's files."""
import os

import torch


class ModelServer:
    def __init__(self, logdir=None, ckpt_period=50):
        self._ckpt_dir = os.path.join(logdir, 'ckpt') if logdir else None
        if self._ckpt_dir:
            os.makedirs(self._ckpt_dir, exist_ok=True)
        self._ckpt_period = ckpt_period
        self._state, self._version = None, 0

    def reset(self):
        self._state, self._version = None, 0

    def update(self, model, version):
        self._state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        self._version = int(version)
        if self._ckpt_dir and self._ckpt_period > 0 and version % self._ckpt_period == 0:
            self.checkpoint(self._ckpt_dir, model, version)

    def latest(self, model):
        if self._state is not None:
            model.load_state_dict(self._state)
        return self._version

    def sync(self, model, src=0, group=None):
        """Multi-process jobs: rank `src` publishes what its last `update` stored, everyone else
        receives it; afterwards `latest` returns the same weights and version on every rank."""
        import torch.distributed as dist

        from .distributed import broadcast_model
        if dist.get_rank(group) == src and self._state is not None:
            model.load_state_dict(self._state)
        self._version = broadcast_model(model, self._version, src=src, group=group)
        self._state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return self._version

    @staticmethod
    def checkpoint(dirname, model, version):
        torch.save({'model_version': version, 'model_state_dict': model.state_dict()},
                   f'{dirname}/{version:06d}.pt')


def load_ckpt(ckpt, model):
    """alphazero/eval.py:135-141 without the config lookup: weights of a checkpoint file into `model`,
    returns its model_version.  Tensors only: nothing else in the file is executed."""
    checkpoint = torch.load(ckpt, map_location='cpu', weights_only=True)
    model.load_state_dict(checkpoint['model_state_dict'])
    return int(checkpoint['model_version'])
