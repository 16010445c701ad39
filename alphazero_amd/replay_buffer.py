"""Replay window with the reference's sample layout (alphazero/replay_buffer.py:105-195).

`ReplayBufferDataset.add_game(game, version)` and `dataset[i]` behave like the reference's
SQLite-backed dataset, but the store is an in-memory list of finished games (histories and
int16 visit counts, exactly what the reference serialises, replay_buffer.py:30-35); the on-disk
SQLite schema is reproduced by `write_sqlite` so existing tooling can read a dump.
"""
import sqlite3

import numpy as np
from torch.utils.data import Dataset, Sampler


class ReplayBufferDataset(Dataset):
    def __init__(self, config, dbpath=None):
        self._config = config
        self._games = []          # (history list, [int16 visits per move], model_version)
        self._states = []         # (game_index, move_index)
        self._dbpath = dbpath

    def __len__(self):
        return len(self._states)

    def add_game(self, game, model_version):
        assert game.terminal
        assert len(game.search_statistics) == len(game.history)
        assert game.search_statistics[0] is not None
        stats = [np.asarray(s).astype(np.int16) for s in game.search_statistics]
        gi = len(self._games)
        self._games.append((list(game.history), stats, int(model_version)))
        self._states.extend((gi, m) for m in range(len(game.history)))

    def num_games(self):
        return len(self._games)

    def num_distinct_games(self):
        return len({tuple(g[0]) for g in self._games})

    def num_states(self):
        return len(self._states)

    @property
    def db(self):
        return self

    def __getitem__(self, i):
        game_index, move_index = self._states[i]
        history, stats, version = self._games[game_index]
        finished = self._config.Game(history=list(history))
        game = self._config.Game(history=list(history[:move_index]))
        valid = game.valid_actions
        n = stats[move_index]
        p = np.zeros(game.NUM_ACTIONS, np.float32)
        p[valid] = n / np.sum(n)
        mask = np.zeros(game.NUM_ACTIONS, bool)
        mask[valid] = True
        z = np.array([finished.terminal_value(move_index % 2)], dtype=np.float32)
        solution = game.solve() if i % 10 == 0 else None
        scores, v_star = np.zeros(game.NUM_ACTIONS, np.float32), 0
        if solution is not None:
            scores[valid], v_star = solution
        return {'x': game.render(), 'z': z, 'p': p, 'p_valid': mask, 'game_index': game_index,
                'move_index': move_index, 'model_version': version, 'solved': solution is not None,
                'v*': v_star, 'move_scores': scores}

    def write_sqlite(self, path):
        """Dump in the reference's schema (replay_buffer.py:15-47)."""
        con = sqlite3.connect(path)
        con.execute('CREATE TABLE IF NOT EXISTS selfplay_games (id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,'
                    ' model_version INTEGER NOT NULL, history TEXT NOT NULL, search_statistics BLOB NOT NULL);')
        con.execute('CREATE TABLE IF NOT EXISTS game_states (id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,'
                    ' game_id INTEGER NOT NULL, move_index INTEGER NOT NULL);')
        for hist, stats, version in self._games:
            cur = con.execute('INSERT INTO selfplay_games (model_version, history, search_statistics) VALUES (?,?,?);',
                              (version, ','.join(str(a) for a in hist), np.concatenate(stats).tobytes()))
            con.executemany('INSERT INTO game_states (game_id, move_index) VALUES (?,?);',
                            [(cur.lastrowid, m) for m in range(len(hist))])
        con.commit()
        con.close()


class ReplayBufferSampler(Sampler):
    """Uniform over the last `bufferlen` states; with replacement only if the batch is larger
    than the window (replay_buffer.py:188-195)."""

    def __init__(self, dataset, batch_size, bufferlen=0):
        self.dataset, self.batch_size, self.bufferlen = dataset, batch_size, bufferlen

    def __len__(self):
        return int(1e9)

    def __iter__(self):
        return self

    def __next__(self):
        hi = len(self.dataset)
        if hi == 0:
            raise StopIteration
        lo = max(hi - self.bufferlen, 0) if self.bufferlen > 0 else 0
        idx = lo + np.random.choice(hi - lo, self.batch_size, self.batch_size > hi - lo)
        return idx.tolist()


class DeviceReplayWindow:
    """The last `capacity` training states, kept on the GPU as tensors (the reference trains from a
    window of 64 Ki states, alphazero/training.py:35-38).  Finished games arrive as raw
    az_game_record bytes (SelfPlayEngine.drain_finished_bytes / distributed.gather_record_bytes) and
    are turned into examples by the az_examples_from_records kernel: the same x, p, z, p_valid and
    model_version that ReplayBufferDataset.__getitem__ returns, without a per-state Python call.
    `sample(batch_size)` draws like ReplayBufferSampler (uniform over the window, with replacement
    only when the batch is larger than the window) but from torch's device RNG."""

    def __init__(self, game_cls, capacity=64 * 1024, device='cuda'):
        import ctypes as C

        import torch

        from . import _lib
        self._torch, self._C, self._lib = torch, C, _lib
        self.lib = _lib.load()
        self.game_id = game_cls.AZ_GAME_ID
        self.A = game_cls.NUM_ACTIONS
        probe = game_cls().render()
        self.shape = tuple(probe.shape)                          # (2, H, W)
        self.capacity, self.device = int(capacity), torch.device(device)
        self.record_bytes = C.sizeof(_lib.GameRecord)
        self.x = torch.zeros((self.capacity,) + self.shape, dtype=torch.float32, device=self.device)
        self.p = torch.zeros(self.capacity, self.A, dtype=torch.float32, device=self.device)
        self.z = torch.zeros(self.capacity, 1, dtype=torch.float32, device=self.device)
        self.p_valid = torch.zeros(self.capacity, self.A, dtype=torch.bool, device=self.device)
        self.model_version = torch.zeros(self.capacity, dtype=torch.int32, device=self.device)
        self.count = 0                                           # states ever added
        self.num_games = 0

    def __len__(self):
        return min(self.count, self.capacity)

    def add_records(self, raw):
        """raw: uint8 [n, sizeof(az_game_record)] (numpy array or CPU tensor).  Returns the number of states added."""
        torch, C = self._torch, self._C
        raw = np.ascontiguousarray(np.asarray(raw, dtype=np.uint8)).reshape(-1, self.record_bytes)
        n = raw.shape[0]
        if n == 0:
            return 0
        off_moves = self._lib.GameRecord.num_moves.offset
        moves = raw[:, off_moves:off_moves + 4].copy().view(np.int32).reshape(n)
        offsets = np.zeros(n + 1, np.int32)
        np.cumsum(moves, out=offsets[1:])
        total = int(offsets[-1])
        recs = torch.from_numpy(raw).to(self.device)
        offs = torch.from_numpy(offsets).to(self.device)
        x = torch.empty((total,) + self.shape, dtype=torch.float32, device=self.device)
        p = torch.empty(total, self.A, dtype=torch.float32, device=self.device)
        z = torch.empty(total, dtype=torch.float32, device=self.device)
        valid = torch.empty(total, self.A, dtype=torch.uint8, device=self.device)
        version = torch.empty(total, dtype=torch.int32, device=self.device)
        ptr = self._lib.ptr
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._lib.check(self.lib.az_examples_from_records(self.game_id, ptr(recs), n, ptr(offs), total, ptr(x), ptr(p),
                                                          ptr(z), ptr(valid), ptr(version), st))
        keep = min(total, self.capacity)                         # a batch larger than the window keeps its tail
        src = slice(total - keep, total)
        dst = (torch.arange(self.count + total - keep, self.count + total, device=self.device) % self.capacity)
        self.x.index_copy_(0, dst, x[src])
        self.p.index_copy_(0, dst, p[src])
        self.z.index_copy_(0, dst, z[src].unsqueeze(1))
        self.p_valid.index_copy_(0, dst, valid[src].bool())
        self.model_version.index_copy_(0, dst, version[src])
        self.count += total
        self.num_games += n
        return total

    def state(self, i):
        """State i of the window, oldest first (for tests and inspection)."""
        j = (self.count - len(self) + i) % self.capacity
        return {'x': self.x[j], 'p': self.p[j], 'z': self.z[j], 'p_valid': self.p_valid[j],
                'model_version': int(self.model_version[j])}

    def sample(self, batch_size, generator=None):
        torch = self._torch
        n = len(self)
        if n == 0:
            raise StopIteration
        if batch_size > n:
            idx = torch.randint(n, (batch_size,), device=self.device, generator=generator)
        else:
            idx = torch.randperm(n, device=self.device, generator=generator)[:batch_size]
        idx = (idx + (self.count - n)) % self.capacity
        return {'x': self.x[idx], 'p': self.p[idx], 'z': self.z[idx], 'p_valid': self.p_valid[idx],
                'model_version': self.model_version[idx]}
