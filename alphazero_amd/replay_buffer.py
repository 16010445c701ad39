"""Replay storage with the reference's sample layout and on-disk format
(alphazero/replay_buffer.py:9-195).

`SelfPlayDatabase` reads and writes the reference's SQLite schema row for row (tables
`selfplay_games` and `game_states`, history as comma-separated text, visit counts as one int16
blob), so databases written by either side open in the other.  `ReplayBufferDataset.add_game` and
`dataset[i]` behave like the reference's dataset; games are kept in memory for sampling and, when
a path is given, also appended to the database (and loaded from it on open).  `RecordSink` moves
the database writes of a self-play loop onto a background thread: it takes the engine's raw
finished-game records and writes them as rows.  `DeviceReplayWindow` is the training window held
on the GPU.
"""
import os
import queue
import sqlite3
import threading

import numpy as np
from torch.utils.data import Dataset, Sampler

_SCHEMA = (
    """CREATE TABLE IF NOT EXISTS selfplay_games (
            id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,
            model_version INTEGER NOT NULL,
            history TEXT NOT NULL,
            search_statistics BLOB NOT NULL);""",
    """CREATE TABLE IF NOT EXISTS game_states (
            id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,
            game_id INTEGER NOT NULL,
            move_index INTEGER NOT NULL);""",
)


def _encode_game(history, stats):
    """(history text, int16 blob) as the reference stores them (replay_buffer.py:30-35); counts
    above 32767 wrap exactly as its astype(int16) does."""
    text = ','.join(str(int(a)) for a in history)
    blob = np.concatenate([np.asarray(s) for s in stats]).astype(np.int16).tobytes()
    return text, blob


class SelfPlayDatabase:
    """The reference's two tables behind its method names (replay_buffer.py:9-87)."""

    def __init__(self, dbpath):
        self._con = sqlite3.connect(dbpath)

    @staticmethod
    def connect(dbpath):
        return SelfPlayDatabase(dbpath)

    def close(self):
        self._con.close()

    def init_db(self):
        for stmt in _SCHEMA:
            self._con.execute(stmt)
        self._con.commit()

    def write_rows(self, rows):
        """rows: iterable of (model_version, history list, per-move visit arrays); one commit."""
        for version, history, stats in rows:
            text, blob = _encode_game(history, stats)
            cur = self._con.execute('INSERT INTO selfplay_games (model_version, history, search_statistics) '
                                    'VALUES (?, ?, ?);', (int(version), text, blob))
            gid = cur.lastrowid
            self._con.executemany('INSERT INTO game_states (game_id, move_index) VALUES (?, ?);',
                                  [(gid, m) for m in range(len(history))])
        self._con.commit()

    def write_game(self, game, model_version):
        assert len(game.search_statistics) == len(game.history)
        assert game.terminal
        assert game.search_statistics[0] is not None
        self.write_rows([(model_version, game.history, game.search_statistics)])

    def read_game(self, i, game):
        assert len(game.history) == 0
        row = self._con.execute('SELECT model_version, history, search_statistics FROM selfplay_games '
                                'WHERE id = ?;', (i + 1,)).fetchone()
        if row is None:
            raise ValueError
        version, text, blob = row
        counts = np.frombuffer(blob, dtype=np.int16)
        n = 0
        for a in (int(t) for t in text.split(',')):
            k = len(game.valid_actions)
            game.take_action(a, counts[n:n + k])
            n += k
        return game, version

    def read_state(self, i):
        row = self._con.execute('SELECT game_id, move_index FROM game_states WHERE id = ?;', (i + 1,)).fetchone()
        if row is None:
            raise ValueError
        return row[0] - 1, row[1]

    def num_games(self):
        return self._con.execute('SELECT COUNT(*) FROM selfplay_games;').fetchone()[0]

    def num_states(self):
        return self._con.execute('SELECT COUNT(*) FROM game_states;').fetchone()[0]

    def num_distinct_games(self):
        return self._con.execute('SELECT COUNT(DISTINCT history) FROM selfplay_games;').fetchone()[0]


class ReplayBufferDataset(Dataset):
    def __init__(self, config, dbpath=None):
        self._config = config
        self._games = []          # (history list, [int16 visits per move], model_version)
        self._states = []         # (game_index, move_index)
        self._dbpath = dbpath
        self._db = None
        if dbpath is not None:
            self._db = SelfPlayDatabase.connect(dbpath)
            self._db.init_db()
            self._pid = os.getpid()
            for i in range(self._db.num_games()):            # an existing database is the window's start
                g, version = self._db.read_game(i, config.Game())
                self._remember(g.history, g.search_statistics, version)

    def _remember(self, history, stats, version):
        gi = len(self._games)
        self._games.append((list(history), [np.asarray(s).astype(np.int16) for s in stats], int(version)))
        self._states.extend((gi, m) for m in range(len(history)))

    def __len__(self):
        return len(self._states)

    def add_game(self, game, model_version):
        assert game.terminal
        assert len(game.search_statistics) == len(game.history)
        assert game.search_statistics[0] is not None
        self._remember(game.history, game.search_statistics, model_version)
        if self._db is not None:
            if os.getpid() != self._pid:                     # SQLite connections do not survive a fork
                self._db, self._pid = SelfPlayDatabase.connect(self._dbpath), os.getpid()
            self._db.write_game(game, model_version)

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
        """Dump every game held in memory to a new database in the reference's schema."""
        db = SelfPlayDatabase.connect(path)
        db.init_db()
        db.write_rows((version, hist, stats) for hist, stats, version in self._games)
        db.close()


class RecordSink:
    """Background writer from raw az_game_record bytes to the reference's SQLite tables, so a
    self-play loop never waits on a commit (the reference commits once per game inside play_game:
    selfplay.py:41, replay_buffer.py:40-47).  `put(raw)` takes the uint8 [n, record size] array of
    SelfPlayEngine.drain_finished_bytes (copied); `close()` drains the queue and joins."""

    def __init__(self, dbpath):
        from ._lib import GameRecord
        self._rec = GameRecord
        self._q = queue.Queue()
        self._error = None
        self.games_written = 0
        self._thread = threading.Thread(target=self._run, args=(dbpath,), daemon=True)
        self._thread.start()

    def _run(self, dbpath):
        try:
            db = SelfPlayDatabase.connect(dbpath)
            db.init_db()
            while True:
                raw = self._q.get()
                if raw is None:
                    break
                rows = []
                for i in range(raw.shape[0]):
                    r = self._rec.from_buffer_copy(raw[i].tobytes())
                    n = r.num_moves
                    rows.append((r.model_version, [int(r.history[m]) for m in range(n)],
                                 [np.array(r.visits[m][:r.num_edges[m]], dtype=np.uint32) for m in range(n)]))
                db.write_rows(rows)
                self.games_written += len(rows)
            db.close()
        except Exception as e:                               # surfaced by put()/close()
            self._error = e

    def put(self, raw):
        if self._error is not None:
            raise self._error
        raw = np.array(raw, dtype=np.uint8, copy=True)
        if raw.shape[0]:
            self._q.put(raw)

    def close(self):
        self._q.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error


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
    only when the batch is larger than the window): from torch's device RNG, or -- given a seed and a draw
    counter in device memory -- in one launch of az_sample_window, which a recorded training step can replay."""

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
        self.count_dev = torch.zeros(1, dtype=torch.int64, device=self.device)    # the same, where kernels read it
        self.num_games = 0

    def __len__(self):
        return min(self.count, self.capacity)

    def add_records(self, raw):
        """raw: uint8 [n, sizeof(az_game_record)] (numpy array or CPU tensor).  Returns the number of states added."""
        raw = np.ascontiguousarray(np.asarray(raw, dtype=np.uint8)).reshape(-1, self.record_bytes)
        if raw.shape[0] == 0:
            return 0
        return self.add_records_device(self._torch.from_numpy(raw).to(self.device))

    def add_records_device(self, recs):
        """recs: uint8 tensor [n, sizeof(az_game_record)] already on the window's device (as
        SelfPlayEngine.take_finished_device / distributed.gather_records_device deliver them): the
        records never visit the host; the only host read is the total number of states."""
        torch, C = self._torch, self._C
        recs = recs.reshape(-1, self.record_bytes).contiguous()
        n = recs.shape[0]
        if n == 0:
            return 0
        off_moves = self._lib.GameRecord.num_moves.offset
        moves = recs[:, off_moves:off_moves + 4].contiguous().view(torch.int32).reshape(n)
        offs = torch.zeros(n + 1, dtype=torch.int32, device=self.device)
        offs[1:] = torch.cumsum(moves, 0)
        total = int(offs[-1].item())
        x = torch.empty((total,) + self.shape, dtype=torch.float32, device=self.device)
        p = torch.empty(total, self.A, dtype=torch.float32, device=self.device)
        z = torch.empty(total, dtype=torch.float32, device=self.device)
        valid = torch.empty(total, self.A, dtype=torch.uint8, device=self.device)
        version = torch.empty(total, dtype=torch.int32, device=self.device)
        ptr = self._lib.ptr
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
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
        self.count_dev.fill_(self.count)
        self.num_games += n
        return total

    def state(self, i):
        """State i of the window, oldest first (for tests and inspection)."""
        j = (self.count - len(self) + i) % self.capacity
        return {'x': self.x[j], 'p': self.p[j], 'z': self.z[j], 'p_valid': self.p_valid[j],
                'model_version': int(self.model_version[j])}

    def _sample_index(self, batch_size, generator=None):
        torch = self._torch
        n = len(self)
        if n == 0:
            raise StopIteration
        if batch_size > n:
            idx = torch.randint(n, (batch_size,), device=self.device, generator=generator)
        else:
            idx = torch.randperm(n, device=self.device, generator=generator)[:batch_size]
        return (idx + (self.count - n)) % self.capacity

    def sample(self, batch_size, generator=None, seed=None, draw_dev=None):
        if seed is not None:
            torch = self._torch
            x = torch.empty((batch_size,) + self.shape, dtype=torch.float32, device=self.device)
            p = torch.empty(batch_size, self.A, dtype=torch.float32, device=self.device)
            z = torch.empty(batch_size, 1, dtype=torch.float32, device=self.device)
            valid = torch.empty(batch_size, self.A, dtype=torch.bool, device=self.device)
            self.sample_into(x, p, z, valid, seed=seed, draw_dev=draw_dev)
            return {'x': x, 'p': p, 'z': z, 'p_valid': valid}
        idx = self._sample_index(batch_size, generator)
        return {'x': self.x[idx], 'p': self.p[idx], 'z': self.z[idx], 'p_valid': self.p_valid[idx],
                'model_version': self.model_version[idx]}

    def kernel_sampling(self, batch_size):
        """True when az_sample_window can draw this batch from this window (its limits: 262144 states, 2048 per batch)."""
        return self.capacity <= (1 << 18) and 0 < batch_size <= 2048

    def sample_into(self, x, p, z, p_valid, generator=None, seed=None, draw_dev=None):
        """The same kind of draw as sample(len(x)), gathered into the caller's tensors (a training step recorded as
        a HIP graph reads its batch from fixed addresses).  With `seed` and `draw_dev` (int64 device tensor [1]) the
        draw and the gather are ONE launch (az_sample_window) that is a function of (seed, draw_dev[0]) and increments
        draw_dev[0]: recorded in a graph, every replay takes a new batch from the window as it then is."""
        torch = self._torch
        if seed is not None:
            C, ptr = self._C, self._lib.ptr
            assert draw_dev is not None and draw_dev.dtype == torch.int64 and draw_dev.device.type == self.device.type
            assert x.is_contiguous() and p.is_contiguous() and z.is_contiguous() and p_valid.is_contiguous()
            if len(self) == 0:
                raise StopIteration
            with torch.cuda.device(self.device):
                st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                self._lib.check(self.lib.az_sample_window(
                    ptr(self.x), ptr(self.p), ptr(self.z), ptr(self.p_valid), int(np.prod(self.shape)), self.A,
                    ptr(self.count_dev), self.capacity, C.c_uint64(int(seed) & (2 ** 64 - 1)), ptr(draw_dev), x.shape[0],
                    ptr(x), ptr(p), ptr(z), ptr(p_valid), st))
            return
        idx = self._sample_index(x.shape[0], generator)
        torch.index_select(self.x, 0, idx, out=x)
        torch.index_select(self.p, 0, idx, out=p)
        torch.index_select(self.z, 0, idx, out=z)
        torch.index_select(self.p_valid, 0, idx, out=p_valid)
