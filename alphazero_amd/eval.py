"""Match play between agents, with the reference's player protocol (alphazero/eval.py:12-132):
`set_game(game)`, `get_action() -> (action, control)`, `observe_action(action)`, and
`play(game, players, show=False)`.  AlphaZeroPlayer searches with the GPU engine through
`run_mcts`; SolverPlayer asks `game.solve()`.  The interactive command line of the reference's
script is not reproduced; `match()` plays a series and counts the outcomes."""
import os

import numpy as np
import torch

from .config import AlphaZeroConfig
from .mcts import MCTreeNode, run_mcts


class Player:
    def set_game(self, game):
        pass

    def get_action(self):
        pass

    def observe_action(self, action):
        pass


class HumanPlayer(Player):
    def __init__(self):
        self._game = None

    def set_game(self, game):
        self._game = game

    def get_action(self):
        g = self._game
        text = input(f'(Player {g.next_player}) Take Action [{g.valid_actions}], (q)uit, (d)ebug: ')
        try:
            return int(text), None
        except ValueError:
            return None, text


class AlphaZeroPlayer(Player):
    """eval.py:40-76: an un-noised search of `num_simulations` per move on a tree that follows the
    game; greedy unless `exploration` (a temperature) is positive."""

    def __init__(self, model, debug=False, num_simulations=128):
        self.exploration = 0
        self.num_simulations = num_simulations
        self._model, self._debug = model, debug
        self._game = self._tree = None

    def set_game(self, game):
        self._game = game
        self._tree = MCTreeNode()
        run_mcts(game, self._tree, self._model, 0, epsilon=0)

    def _advance(self, action_index):
        self._tree = self._tree.traverse(action_index)
        self._tree.kill_siblings()

    def get_action(self):
        if self._debug:
            print(self._game)
        run_mcts(self._game, self._tree, self._model, self.num_simulations, epsilon=0)
        if self._debug:
            n = self._tree.num_visits
            print(f'Num visits ({np.sum(n)}): {n}')
            print(f'Action Prior: {self._tree.action_prior}')
        i = self._tree.sample_action(self.exploration) if self.exploration > 0 else self._tree.greedy_action()
        self._advance(i)
        return self._game.valid_actions[i], None

    def observe_action(self, action):
        self._advance(self._game.valid_actions.index(action))


class SolverPlayer(Player):
    """eval.py:79-93: a uniformly random move among the solver's best."""

    def __init__(self):
        self._game = None

    def set_game(self, game):
        self._game = game

    def get_action(self):
        scores, _ = self._game.solve()
        best = np.flatnonzero(np.asarray(scores) == np.max(scores))
        return self._game.valid_actions[int(np.random.choice(best))], None


def play(game, players, show=False):
    """Play `game` to the end from its current state (eval.py:96-132).  Control strings from a
    player: 'q' stops the match (returns False), anything else asks again."""
    for p in players:
        p.set_game(game)
        show = show or isinstance(p, HumanPlayer)
    while not game.terminal:
        player = players[game.next_player]
        if show:
            print(game)
        action, control = player.get_action()
        if control is not None:
            if control == 'q':
                return False
            if control == 'd':
                import pdb
                pdb.set_trace()
            else:
                print('Invalid Selection.')
            continue
        if action not in game.valid_actions:
            print('Invalid Action.')
            continue
        for p in players:
            if p is not player:
                p.observe_action(action)
        game.take_action(action)
    if show:
        print(game)
        print(f'Player {game.winner} wins!' if game.winner >= 0 else 'Game drawn!')
    return True


def match(game_cls, players, num_games):
    """`num_games` games from the empty board, sides alternating; (wins of players[0], draws, wins of players[1])."""
    tally = [0, 0, 0]
    for i in range(num_games):
        order = players if i % 2 == 0 else players[::-1]
        game = game_cls()
        play(game, order)
        w = game.winner
        tally[1 if w < 0 else (0 if order[w] is players[0] else 2)] += 1
    return tuple(tally)


def load_ckpt(ckpt, device='cpu'):
    """(model, config) from a checkpoint written by ModelServer.checkpoint or by the reference
    (eval.py:135-141): the run's config.yaml sits one directory above the checkpoint's."""
    config = AlphaZeroConfig(os.path.join(os.path.dirname(ckpt), '../config.yaml'))
    model = config.Model()
    state = torch.load(ckpt, map_location=device, weights_only=True)
    model.load_state_dict(state['model_state_dict'])
    return model, config
