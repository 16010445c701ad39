"""Two-player alternating game over an action history.

Host-side mirror of the reference's `Game` protocol (alphazero/game.py:5-65): the
same attributes and methods, so objects can be handed to code written against
the reference (replay buffers, evaluation scripts).  Rule queries on a single
object are answered on the host from bitboards; the self-play hot path never
goes through these objects -- it runs G games at once inside the HIP engine and
only materialises `Game` objects for finished games.
"""
import copy


class Game:
    NUM_ACTIONS = -1
    AZ_GAME_ID = None      # engine game id for built-in games, None for user games

    def __init__(self, history=None, search_statistics=None):
        assert self.NUM_ACTIONS > 0, 'Game classes must override NUM_ACTIONS class attribute.'
        self.history = list(history) if history else []
        self.search_statistics = search_statistics or [None] * len(self.history)
        assert len(self.history) == len(self.search_statistics), \
            'Length of game and search statistics must match.'

    def __len__(self):
        return len(self.history)

    def __str__(self):
        raise NotImplementedError

    @property
    def next_player(self):
        return len(self.history) & 1

    @property
    def prev_player(self):
        return 1 - self.next_player

    def take_action(self, action, search_statistics=None):
        assert action in self.valid_actions, f'Cannot take invalid action "{action}"'
        self.history.append(action)
        self.search_statistics.append(search_statistics)

    def clone(self):
        return copy.deepcopy(self)

    @property
    def terminal(self):
        raise NotImplementedError

    def terminal_value(self, player):
        w = self.winner
        assert w is not None
        if w < 0:
            return 0
        return 1 if player == w else -1

    @property
    def winner(self):
        raise NotImplementedError

    @property
    def valid_actions(self):
        raise NotImplementedError

    @property
    def action_names(self):
        return [str(a) for a in self.valid_actions]

    def render(self, turn=None):
        raise NotImplementedError

    def solve(self):
        return None

    # ---- engine interop (built-in games) ---------------------------------
    def position(self):
        """(stones of side to move, stones of previous mover, moves played) as bitboards."""
        raise NotImplementedError
