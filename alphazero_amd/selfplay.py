"""SelfPlayWorker with the reference's constructor and `play_game()` contract
(alphazero/selfplay.py:5-42), running `num_games` games concurrently on one GPU.

The reference plays one game per call; here one call advances every game slot in
lock-step, move by move, until at least one game has finished, hands each finished
game to `dataset.add_game(game, model_version)` as a `Game` object carrying the
history and per-move visit counts, and returns the model version -- so the loops in
the reference's train.py (`while version < num_steps: worker.play_game()`) work as
they are.  Collaborators are duck-typed exactly as in the reference:
`model_server.latest(model) -> int`, `dataset.add_game(game, version)`,
`config.Game()`, `config.Model()`, `config.selfplay[...]`.

Where this differs from the reference, because game slots outlive a call:
  * by default a game can span several model versions (weights are refreshed once per `play_game()` /
    `play_moves()` call, games finish whenever they finish), and with `reuse_search_tree` its tree keeps
    the priors and values of the weights that expanded each node -- including the uniform prior of
    version 0 across the switch to the first real weights.  The reference plays a whole game with one
    version (selfplay.py:18,41).  `dataset.add_game(game, v)` gets the version current when the game
    ENDED; the record also carries the version current when it STARTED (`first_model_version`).
    `one_version_per_game=True` gives the reference's rule instead: when `refresh_model()` finds new weights the
    games in progress are given up (az_abandon_games) and `play_game()` plays a whole generation -- until every
    slot has finished the game it started under these weights -- so no game and no search tree ever sees two
    versions, at the price of the games that were under way when the weights changed.
  * pool exhaustion and record-ring overflow cannot happen in the reference; here `collect_finished*`
    raises AzError as soon as the engine reports either.
"""
import numpy as np
import torch

from . import engine as _engine
from .mcts import evaluate_leaves
from .models import UniformModel


def record_to_game(game_cls, rec):
    """az_game_record -> Game with history and compacted uint32 visit counts per move."""
    n = rec.num_moves
    hist = [int(rec.history[i]) for i in range(n)]
    stats = [np.array(rec.visits[i][:rec.num_edges[i]], dtype=np.uint32) for i in range(n)]
    return game_cls(history=hist, search_statistics=stats)


class SelfPlayWorker:
    def __init__(self, config, model_server, dataset, device, num_games=1024, seed=0,
                 rank=0, world_size=1, evaluator=None, eval_cache_log2=0, net_precision='bf16',
                 one_version_per_game=False, streams=1):
        # net_precision (Connect Four networks with up to 32 trunk channels): 'bf16' = the bf16 MFMA network, the
        # fast path; 'fp32' = the network at the reference's own precision (alphazero/mcts.py:94-107 evaluates in
        # fp32) on fp16-pair operands, outputs within 1e-5 of the torch module, about a third of the bf16 rate
        # streams = 2 (networks evaluated stepwise by a HIP kernel of this package, e.g. the 128-channel one): the games
        # are split over two engines on two HIP streams, so that one half's search step and small kernels run under
        # the other half's network evaluation; the halves own the game uids of the one-engine run, so the same games
        # are played
        if net_precision not in ('bf16', 'fp32'):
            raise ValueError("net_precision is 'bf16' or 'fp32'")
        if streams not in (1, 2) or (streams == 2 and (evaluator is not None or int(num_games) % 2)):
            raise ValueError('streams is 1 or 2; two streams need an even number of games and a built-in network')
        if evaluator is not None and eval_cache_log2:
            # az_sim_step answers cached leaves by writing rows of the (p, v) buffers it is given; a
            # caller's evaluator may return fresh tensors or leave other rows undefined, which would feed
            # garbage into expand/backup without any error.  The built-in HIP networks own persistent
            # buffers and touch only the rows they evaluate, so the cache is theirs alone.
            raise ValueError('eval_cache_log2 works with the built-in HIP networks only, not with a '
                             'caller-supplied evaluator')
        self._config = config
        self._model_server = model_server
        self._dataset = dataset
        self._device = torch.device(device)
        self._model = config.Model().to(self._device)
        self._game_cls = type(config.Game())
        sp = config.selfplay
        self.num_games = int(num_games)
        self.num_simulations = int(sp['num_expansions_per_sim'])
        dev_index = self._device.index if self._device.index is not None else torch.cuda.current_device()

        def make_engine(games, first_uid):
            return _engine.SelfPlayEngine(
                self._game_cls.AZ_GAME_ID, games, self.num_simulations,
                exploration_depth=sp['exploration_depth'], reuse_search_tree=sp['reuse_search_tree'],
                seed=seed, device=dev_index, first_game_uid=first_uid, uid_stride=world_size * self.num_games)
        self.engines = [make_engine(self.num_games // streams, rank * self.num_games)]
        self._evaluator = evaluator           # callable(leaf_bb, leaf_info) -> (p, v); None = built-in
        self._nets = [None]                   # the HIP network of each half (one half unless streams=2)
        if evaluator is None:
            # the shipped nets run in hand-written HIP kernels: Connect Four with up to 32 trunk
            # channels in the one-launch fused kernel and with 128 channels in the stem / tower / heads
            # kernels, tic-tac-toe (4, 8 or 16 channels) in its own fp32 kernel; anything else is
            # evaluated by calling the torch module on the rendered leaves
            from .models import AlphaZero, AlphaZeroC4
            from .net import CMAX, C4Net128, FusedC4Net, FusedC4NetHX, TTTNetF32
            if isinstance(self._model, AlphaZeroC4) and self._model.block_channels <= CMAX:
                self._fused = (FusedC4NetHX(self._model, self._device) if net_precision == 'fp32'
                               else FusedC4Net.from_module(self._model, self._device))
            elif isinstance(self._model, AlphaZeroC4) and self._model.block_channels == 128:
                self._fused = C4Net128(self._model, self._device, max_leaves=self.num_games // streams)
            elif (type(self._model) is AlphaZero and self._game_cls.AZ_GAME_ID == _engine._lib.GAME_TTT
                  and TTTNetF32.supports(self._model)):
                self._fused = TTTNetF32(self._model, self._device)
        self._streams = None
        if streams == 2:
            from .net import FusedC4Net as _F0, FusedC4NetHX as _FH0
            if self._fused is None or isinstance(self._fused, (_F0, _FH0)):
                # (the <= 32-channel Connect Four networks run a whole move in one launch: nothing to overlap)
                raise ValueError('streams=2 is for networks evaluated stepwise by a HIP kernel (128-channel Connect Four, '
                                 'tic-tac-toe)')
            half = self.num_games // 2
            self.engines.append(make_engine(half, rank * self.num_games + half))
            self._nets.append(type(self._fused)(self._model, self._device, max_leaves=half)
                              if isinstance(self._fused, C4Net128) else type(self._fused)(self._model, self._device))
            torch.cuda.synchronize(self._device)       # buffers made on the current stream, used on the halves' streams
            self._streams = [torch.cuda.Stream(device=self._device) for _ in range(2)]
        if eval_cache_log2:
            # leaves whose position was already evaluated under the current model version are answered
            # by the engine; sound only for evaluators that are pure, batch-independent functions of
            # the position, which the HIP networks are and a torch module in general is not
            if self._fused is None:
                raise ValueError('eval_cache_log2 needs one of the built-in HIP networks')
            for eng in self.engines:
                eng.enable_eval_cache(eval_cache_log2)
        # Connect Four with the <= 32-channel network: a whole move's simulations run in one launch
        # (az_c4_move_fused)
        from .net import FusedC4Net as _F, FusedC4NetHX as _FH
        self._fused_move = (isinstance(self._fused, (_F, _FH)) and evaluator is None
                            and self.num_games <= self.engine.fused_max_games())
        if self._fused is not None and not self._fused_move and self.num_games > 32768:
            raise ValueError('the HIP network kernels take at most 32768 leaves per call: num_games <= 32768')
        self._version = 0
        self._one_version = bool(one_version_per_game)
        self._first_uid, self._uid_stride = rank * self.num_games, world_size * self.num_games
        self.positions = 0
        self.games_finished = 0
        self.games_abandoned = 0

    @property
    def engine(self):
        """The engine (the first half's when the games are split over two)."""
        return self.engines[0]

    @engine.setter
    def engine(self, eng):
        self.engines[0] = eng

    @property
    def _fused(self):
        return self._nets[0]

    @_fused.setter
    def _fused(self, net):
        self._nets[0] = net

    # ---- one move for every game slot -----------------------------------------------
    def _eval(self, bb, info, h=0):
        if self._evaluator is not None:
            return self._evaluator(bb, info)
        if self._nets[h] is not None:
            return self._nets[h](bb, info)
        return evaluate_leaves(self.engine.game, self._model, bb, info)

    def _halves(self):
        """(index, engine) of each half, each inside its own stream when there are two."""
        for h, eng in enumerate(self.engines):
            if self._streams is None:
                yield h, eng
            else:
                with torch.cuda.stream(self._streams[h]):
                    yield h, eng

    def play_move(self, uniform):
        S = self.num_simulations
        if uniform:
            # UniformModel (selfplay.py:19-22) is evaluated inside the search kernel
            for h, eng in self._halves():
                eng.begin_move_uniform()
                eng.run_sims_uniform(S)
        elif S > 0 and self._fused_move and self._fused is not None:
            eng = self.engine
            bb, info = eng.begin_move()
            p, _ = self._eval(bb, info)
            eng.apply_root(p)
            eng.move_fused(self._fused)                # all S simulations, search and network, in one launch
        else:
            leaves = [None] * len(self.engines)
            for h, eng in self._halves():
                bb, info = eng.begin_move()
                p, _ = self._eval(bb, info, h)
                eng.apply_root(p)
                if S > 0:
                    leaves[h] = eng.select()
            # the halves take turns simulation by simulation: each stream always has the other half's launches to
            # run under its network evaluation
            for k in range(S):
                for h, eng in self._halves():
                    p, v = self._eval(*leaves[h], h)
                    if k < S - 1:
                        leaves[h] = eng.sim_step(p, v)
                    else:
                        eng.expand_backup(p, v)
        for h, eng in self._halves():
            eng.choose_and_advance()
        self.positions += self.num_games

    def _check_engine(self):
        flags = 0
        for eng in self.engines:
            flags |= eng.error_flags()
        if flags:
            raise _engine.AzError(
                ('a game ran out of node-pool space (raise node_capacity); ' if flags & 1 else '') +
                ('finished games were dropped, the record ring was full (collect more often); ' if flags & 2 else '') +
                ('a launch of the one-launch-per-move kernel was abandoned by its watchdog' if flags & 16 else ''))

    def collect_finished(self, slots_done=None):
        recs = []
        for h, eng in self._halves():
            recs.extend(eng.drain_finished())
        self._check_engine()
        for r in recs:
            self._dataset.add_game(record_to_game(self._game_cls, r), int(r.model_version))
            if slots_done is not None:
                slots_done[(int(r.uid) - self._first_uid) % self._uid_stride] = True
        self.games_finished += len(recs)
        return len(recs)

    def collect_finished_bytes(self):
        """Finished games as raw az_game_record bytes (uint8 [n, record size]); `dataset` is not
        touched.  For loops that feed a DeviceReplayWindow or ship records to another rank."""
        raw = np.concatenate([eng.drain_finished_bytes() for h, eng in self._halves()])
        self._check_engine()
        self.games_finished += raw.shape[0]
        return raw

    def collect_finished_device(self):
        """Finished games left on the device: uint8 tensor [n, record size] (a copy the caller owns);
        one 8-byte host read for the count and the error bits."""
        out = []
        for h, eng in self._halves():
            recs, head = eng.take_finished_device()
            n, flags = head.tolist()
            self.engine_flags = flags
            if flags:
                raise _engine.AzError(f'engine error bits {flags}: 1 = node pool exhausted, 2 = finished games dropped, '
                                      '16 = a one-launch move was abandoned by its watchdog')
            self.games_finished += n
            out.append(recs[:n].clone())
        if self._streams is not None:
            torch.cuda.synchronize(self._device)       # the caller reads them on its own stream
        return out[0] if len(out) == 1 else torch.cat(out)

    def take_finished_device(self):
        """The engine's hand-over as it is -- (records uint8 [ring capacity, record size], head int32[2] =
        {count, error bits}), both still on the device and not read here -- for callers that ship them to
        another rank first (distributed.gather_records_device) and report what came of it with
        `account_finished`."""
        if len(self.engines) != 1:
            raise ValueError('take_finished_device hands over ONE engine\'s ring: use collect_finished_device with streams=2')
        return self.engine.take_finished_device()

    def account_finished(self, count, flags):
        """Book-keeping for hand-overs whose head was read elsewhere: raises on engine error bits like the
        collect_* methods, then counts the games."""
        self.engine_flags = int(flags)
        if flags:
            raise _engine.AzError(f'engine error bits {flags}: 1 = node pool exhausted, 2 = finished games dropped, '
                                  '8 = visit sum beyond the PUCT table, 16 = a one-launch move was abandoned by its watchdog')
        self.games_finished += int(count)

    def refresh_model(self):
        """selfplay.py:16-22: pull the latest weights; version 0 means 'no model yet'."""
        before = self._version
        self._version = int(self._model_server.latest(self._model))
        for h, eng in self._halves():
            eng.set_model_version(self._version)
            if self._nets[h] is not None and self._version > 0:
                self._nets[h].load_module(self._model)
        if self._one_version and self._version != before:
            # selfplay.py:16-22,41: a game is played from start to end with the weights that were the latest when it
            # began; the games under way belong to the old weights
            if self.positions:
                self.collect_finished()
                for h, eng in self._halves():
                    eng.abandon_games()
                self.games_abandoned += self.num_games
            else:
                for h, eng in self._halves():
                    eng.reset_games()                  # nothing played yet: the empty games just take the version
        return self._version

    def play_game(self):
        version = self.refresh_model()
        uniform = version <= 0 and self._evaluator is None
        if self._one_version:
            # a generation: every slot finishes the game it has under way (all of them under `version`; a slot that
            # finishes early starts another one under the same weights, which is collected if it ends in time)
            done = np.zeros(self.num_games, dtype=bool)
            while not done.all():
                self.play_move(uniform)
                self.collect_finished(done)
            return version
        finished = 0
        while finished == 0:
            self.play_move(uniform)
            finished = self.collect_finished()
        return version

    def play_moves(self, num_moves):
        """Throughput-oriented loop: `num_moves` lock-step moves, finished games collected at the end."""
        version = self.refresh_model()
        uniform = version <= 0 and self._evaluator is None
        for _ in range(num_moves):
            self.play_move(uniform)
            self.collect_finished()       # the ring holds 2*G records; one small D2H per move
        self.collect_finished()
        return version
