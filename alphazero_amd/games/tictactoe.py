"""Tic-tac-toe (interface of alphazero/games/tictactoe.py:9-72); squares 0..8 row-major."""
import numpy as np

from ..game import Game

_LINES = (0o007, 0o070, 0o700, 0o111, 0o222, 0o444, 0o421, 0o124)


def _line(b):
    return any((b & m) == m for m in _LINES)


_SOLVED = {}


def _minimax(p0, p1, root_player):
    """Game-theoretic score of the position for `root_player`, with the reference's scale
    (alphazero/games/tictactoe.py:87-111): outcome (+1, 0, -1) times 5 - moves // 2, so quicker wins
    and slower losses score higher.  Returns (score, scores of the legal moves in ascending order)."""
    key = (p0, p1, root_player)
    hit = _SOLVED.get(key)
    if hit is not None:
        return hit
    n = bin(p0 | p1).count('1')
    won0, won1 = _line(p0), _line(p1)
    if n >= 5 and (won0 or won1 or n == 9):
        w = 0 if won0 else 1 if won1 else -1
        res = ((5 - n // 2) * (0 if w < 0 else 1 if w == root_player else -1), [])
    else:
        mover = n & 1
        scores = []
        for a in range(9):
            if not ((p0 | p1) >> a) & 1:
                scores.append(_minimax(p0 | (1 << a), p1, root_player)[0] if mover == 0 else
                              _minimax(p0, p1 | (1 << a), root_player)[0])
        res = (max(scores) if mover == root_player else min(scores), scores)
    _SOLVED[key] = res
    return res


class TicTacToe(Game):
    NUM_ACTIONS = 9
    AZ_GAME_ID = 0
    BOARD_SHAPE = (3, 3)

    def position(self, turn=None):
        hist = self.history if turn is None else self.history[:turn]
        nx = pv = 0
        for a in hist:
            nx, pv = pv, nx | (1 << a)
        return nx, pv, len(hist)

    @property
    def valid_actions(self):
        nx, pv, _ = self.position()
        return [a for a in range(9) if not ((nx | pv) >> a) & 1]

    @property
    def winner(self):
        n = len(self.history)
        if n < 5:
            return None
        nx, pv, _ = self.position()
        p0, p1 = (pv, nx) if n & 1 else (nx, pv)
        if _line(p0):                      # player 0's lines are looked at first
            return 0
        if _line(p1):
            return 1
        if n == 9:
            return -1
        return None

    @property
    def terminal(self):
        n = len(self.history)
        if n < 5:
            return False
        return n == 9 or self.winner is not None

    def render(self, turn=None):
        if turn is not None:
            assert 0 <= turn <= len(self.history)
            hist = self.history[:turn]
        else:
            hist = self.history
        img = np.zeros((2, 9), dtype=np.float32)
        # plane 0: squares of the player to move in the FULL game, as the reference indexes them
        img[0, hist[self.next_player::2]] = 1
        img[1, hist[self.prev_player::2]] = 1
        return img.reshape(2, 3, 3)

    def solve(self):
        """(score of every legal move, game value in {-1, 0, 1}) for the player to move
        (alphazero/games/tictactoe.py:74-117)."""
        n = len(self.history)
        nx, pv, _ = self.position()
        p0, p1 = (pv, nx) if n & 1 else (nx, pv)
        s, scores = _minimax(p0, p1, n & 1)
        return list(scores), (1 if s > 0 else -1 if s < 0 else 0)

    def __str__(self):
        h = self.history
        t = [' ' if i not in h else 'XO'[h.index(i) % 2] for i in range(9)]
        bar = '\n' + '-' * 11 + '\n'
        return bar.join(' ' + ' | '.join(t[r:r + 3]) for r in (0, 3, 6))
