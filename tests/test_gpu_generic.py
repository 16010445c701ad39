"""run_mcts on Game subclasses the device has no rules for: the tree lives on the engine, the rules stay
with the caller's object (az_generic_*).  The reference's own chain-game test of value back-propagation
(alphazero/test/test_mcts.py:69-180) restated against this package, and the generic path cross-checked bit
for bit against the built-in Connect Four / tic-tac-toe paths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# a one-action game that player 0 wins on the fifth move; the "network" knows the values of the line
CHAIN = {
    (): {'v': 0.9, 'p': [1.]},
    (0,): {'v': -0.9, 'p': [1.]},
    (0, 0): {'v': 0.9, 'p': [1.]},
    (0, 0, 0): {'v': -0.9, 'p': [1.]},
    (0, 0, 0, 0): {'v': 0.9, 'p': [1.]},
    (0, 0, 0, 0, 0): {'winner': 0},
}


class ChainModel:
    def __init__(self, spec):
        self._spec = spec

    def eval(self):
        pass

    def __call__(self, x, p_valid):
        history = tuple(x.int().tolist()[0])
        return torch.tensor([self._spec[history]['p']]), torch.tensor([self._spec[history]['v']])


def _chain_game():
    from alphazero_amd.game import Game

    class ChainGame(Game):
        def __init__(self, spec):
            self.NUM_ACTIONS = len(spec[()]['p'])
            super().__init__()
            self._spec = spec

        def __str__(self):
            return ''

        @property
        def terminal(self):
            return self.winner is not None

        @property
        def winner(self):
            return self._spec[tuple(self.history)].get('winner')

        @property
        def valid_actions(self):
            return list(range(self.NUM_ACTIONS))

        def render(self, turn=None):
            return np.array(self.history, dtype=np.float32)
    return ChainGame(CHAIN)


def test_value_backpropagation_on_a_user_defined_game():
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    model, game, root = ChainModel(CHAIN), _chain_game(), MCTreeNode()
    prev_root = prev_child = None
    for i in range(20):
        run_mcts(game, root, model, 1)
        assert int(root.num_visits.sum()) == i + 1
        q = float(root.mean_value[0])
        if i < 4:
            assert abs(q - 0.9) < 1e-6                      # the first four simulations back 0.9 up to the root
        else:
            assert q > prev_root                            # then player 0's win starts to show
        prev_root = q
        if i > 0:
            qc = float(root.children[0].mean_value[0])
            if i < 4:
                assert abs(qc + 0.9) < 1e-6
            else:
                assert qc < prev_child
            prev_child = qc
    # searching from a child of an existing tree must leave the ancestors alone
    child = root.children[0]
    game.take_action(0)
    run_mcts(game, child, model, 1)
    assert abs(float(root.mean_value[0]) - prev_root) < 1e-6
    assert float(child.mean_value[0]) < prev_child
    assert child.parent is root


@pytest.mark.parametrize('name', ['c4', 'ttt'])
def test_generic_path_equals_the_built_in_rules(name):
    """The same seeded searches through the device rules and through the caller's Game object: identical
    visit counts, W and P bits, tree shapes, with tree re-use across moves."""
    from alphazero_amd.games import BY_NAME
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.models import UniformModel
    cls = BY_NAME[name]
    plain = type('Plain' + cls.__name__, (cls,), {'AZ_GAME_ID': None})      # same rules, unknown to the device

    def dump(node, path=()):
        out = []
        if node is None or not node.expanded:
            return out
        out.append((path, node._edges.tobytes()))
        for i, c in enumerate(node.children):
            out.extend(dump(c, path + (i,)))
        return out

    trees = []
    for game_cls in (cls, plain):
        np.random.seed(3)
        game, tree, snaps = game_cls(), MCTreeNode(), []
        for mv in range(5):
            run_mcts(game, tree, UniformModel(), 40)
            snaps.append(dump(tree))
            ai = tree.greedy_action()
            game.take_action(game.valid_actions[ai], search_statistics=tree.num_visits)
            tree = tree.traverse(ai)
            tree.kill_siblings()
        trees.append((snaps, list(game.history)))
    assert trees[0][1] == trees[1][1]
    for a, b in zip(trees[0][0], trees[1][0]):
        assert a == b


def test_generic_game_with_too_many_actions_is_refused():
    from alphazero_amd.game import Game
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.models import UniformModel

    class Wide(Game):
        NUM_ACTIONS = 16

        @property
        def valid_actions(self):
            return list(range(16))
    with pytest.raises(TypeError, match='15 actions'):
        run_mcts(Wide(), MCTreeNode(), UniformModel(), 1)


def test_search_path_deeper_than_the_stack_is_refused_not_corrupted():
    """A one-action game of 60 moves: once the tree under the root is a chain of more than 48 expanded nodes the
    next select cannot record its path.  run_mcts must raise (error bit 5) and leave the tree as it was."""
    from alphazero_amd import _lib
    from alphazero_amd.game import Game
    from alphazero_amd.mcts import MCTreeNode, run_mcts

    class LongChain(Game):
        NUM_ACTIONS = 1

        def __str__(self):
            return ''

        @property
        def terminal(self):
            return len(self.history) >= 60

        @property
        def winner(self):
            return 0 if self.terminal else None

        @property
        def valid_actions(self):
            return [0]

        def render(self, turn=None):
            return np.zeros((2, 1, 1), np.float32)

    class Flat:
        def eval(self):
            pass

        def __call__(self, x, p_valid):
            return torch.zeros(1, 1), torch.zeros(1, 1)

    game, root = LongChain(), MCTreeNode()
    run_mcts(game, root, Flat(), 47, epsilon=0.0)             # root + 47 nodes: the deepest select has 48 levels
    assert int(root.num_visits.sum()) == 47
    run_mcts(game, root, Flat(), 1, epsilon=0.0)              # 48 levels: still fits
    before = int(root.num_visits.sum())
    with pytest.raises(_lib.AzError, match='48 levels'):
        run_mcts(game, root, Flat(), 1, epsilon=0.0)
    assert int(root.num_visits.sum()) == before
    # nearer the leaves the same tree is searchable again
    node = root
    for _ in range(10):
        game.take_action(0)
        node = node.children[0]
    run_mcts(game, node, Flat(), 3, epsilon=0.0)
    assert int(node.num_visits.sum()) == before - 10 + 3
