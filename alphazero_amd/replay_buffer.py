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
