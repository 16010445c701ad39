"""Connect Four, 7 columns x 6 rows (interface of alphazero/games/connectfour.py:75-139).

Bitboard layout shared with the HIP engine: bit (col*7 + row), row 0 at the
bottom; the 7th bit of every column is a guard bit.
"""
import numpy as np

from ..game import Game

GRID_WIDTH, GRID_HEIGHT, STREAK_LEN = 7, 6, 4
N_SPACES = GRID_WIDTH * GRID_HEIGHT


def _four(b):
    for sh in (7, 6, 8, 1):
        t = b & (b >> sh)
        if t & (t >> (2 * sh)):
            return True
    return False


class ConnectFour(Game):
    NUM_ACTIONS = GRID_WIDTH
    AZ_GAME_ID = 1
    BOARD_SHAPE = (GRID_HEIGHT, GRID_WIDTH)

    def position(self):
        nx = pv = 0
        for a in self.history:
            m = nx | pv
            stone = (m + (1 << (a * 7))) & (0x3F << (a * 7))
            nx, pv = pv, nx | stone
        return nx, pv, len(self.history)

    @property
    def valid_actions(self):
        nx, pv, _ = self.position()
        m = nx | pv
        return [c for c in range(GRID_WIDTH) if not (m >> (c * 7 + 5)) & 1]

    @property
    def winner(self):
        n = len(self.history)
        if n < 2 * STREAK_LEN - 1:
            return None
        _, pv, _ = self.position()
        if _four(pv):                      # only the previous mover can have won
            return self.prev_player
        if n == N_SPACES:
            return -1
        return None

    @property
    def terminal(self):
        n = len(self.history)
        if n < 2 * STREAK_LEN - 1:
            return False
        return n == N_SPACES or self.winner is not None

    def render(self, turn=None):
        nx, pv, _ = self.position()
        board = np.zeros((2, GRID_HEIGHT, GRID_WIDTH), dtype=np.float32)
        for plane, b in enumerate((nx, pv)):
            for c in range(GRID_WIDTH):
                col = (b >> (c * 7)) & 0x3F
                for r in range(GRID_HEIGHT):
                    if (col >> r) & 1:
                        board[plane, r, c] = 1
        return board

    def __str__(self):
        board = self.render()
        mark = ('X', 'O') if self.next_player == 0 else ('O', 'X')
        rows = []
        for r in reversed(range(GRID_HEIGHT)):
            cells = [mark[0] if board[0, r, c] else mark[1] if board[1, r, c] else ' '
                     for c in range(GRID_WIDTH)]
            rows.append('|' + '|'.join(cells) + '|')
        rows.append(' ' + ' '.join(str(i) for i in range(GRID_WIDTH)))
        return '\n'.join(rows)
