"""Self-play and training in one loop, the shape of the reference's `_synchronous_play_and_train`
(alphazero/train.py:85-95), for one GPU or one process per GPU:

    every rank : SelfPlayWorker plays lock-step moves for its shard of the games
    records    : raw az_game_record bytes, device to device: engine ring -> gather on the trainer rank (RCCL)
                 -> DeviceReplayWindow, never through host memory -- replaces the SQLite file
    trainer    : DeviceReplayWindow.add_records_device -> TrainingWorker.process_batch (one SGD step)
    weights    : ModelServer.update on the trainer rank, ModelServer.sync (one broadcast) everywhere

With data_parallel=True every rank trains instead: records are all-gathered, every rank keeps the same
replay window and takes batch_size / world_size samples of its own per step, gradients are averaged
with one all-reduce (distributed.allreduce_gradients) and the ranks' weights stay identical without
any broadcast.

Nothing here is on the measured hot path; it is the caller on either side of it."""
import torch
import torch.distributed as dist

from .model_server import ModelServer
from .replay_buffer import DeviceReplayWindow
from .selfplay import SelfPlayWorker
from .training import TrainingWorker


def synchronous_play_and_train(config, device, num_games=1024, num_steps=None, moves_per_step=1, logdir=None,
                               summary_writer=None, seed=0, trainer_rank=0, window_states=64 * 1024, log=None,
                               data_parallel=False, pipelined=False):
    """Runs until the model version reaches `num_steps` (default: config.training['num_steps']).
    Returns (model_server, selfplay_worker, training_worker or None).
    pipelined (one process only): the training step is queued on the device behind the move's kernels BEFORE the host
    waits for that move's finished games, so the device never idles between the two (an idle device drops its
    clocks and a 1.5 ms step then takes 5-10 ms); the step trains on the window as it was one move earlier."""
    multi = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if multi else 0
    world = dist.get_world_size() if multi else 1
    num_steps = int(num_steps if num_steps is not None else config.training['num_steps'])
    server = ModelServer(logdir if rank == trainer_rank else None)
    server.reset()
    worker = SelfPlayWorker(config, server, None, device, num_games=num_games, seed=seed, rank=rank, world_size=world)
    data_parallel = bool(data_parallel and multi)      # then every rank's own store holds the same weights
    trainer = window = None
    if rank == trainer_rank or data_parallel:
        from .distributed import allreduce_gradients
        window = DeviceReplayWindow(type(config.Game()), window_states, device)
        if data_parallel:
            share = config.training['batch_size'] // world
            assert share * world == config.training['batch_size'], 'batch_size must divide by the number of ranks'
            torch.manual_seed(seed)                        # identical initial weights on every rank ...
            trainer = TrainingWorker(config, server, window, summary_writer if rank == trainer_rank else None, device,
                                     grad_sync=allreduce_gradients, batch_size=share)
            torch.manual_seed(seed + 1 + rank)             # ... and a different sample stream per rank
        else:
            trainer = TrainingWorker(config, server, window, summary_writer, device)
    version = 0
    while version < num_steps:
        # alphazero/selfplay.py:16-22: newest weights first (version 0 = UniformModel)
        if multi and not data_parallel:
            server.sync(worker._model, src=trainer_rank)
        worker.refresh_model()
        uniform = worker._version <= 0
        for _ in range(moves_per_step):
            worker.play_move(uniform)
        if multi and data_parallel:
            # every rank keeps the whole window: records of all ranks on all ranks
            from .distributed import gather_record_bytes
            raw = gather_record_bytes(worker.collect_finished_bytes(), dst=None, device=device)
            window.add_records(raw)
        elif multi:
            # records stay in device memory from the engine's ring to the trainer's window
            from .distributed import gather_records_device
            recs, head = worker.take_finished_device()
            recs, flags, mine = gather_records_device(recs, head, dst=trainer_rank)
            worker.account_finished(mine, flags)           # raises on any rank's error bits, counts this rank's games
            if trainer is not None:
                window.add_records_device(recs)
        elif pipelined:
            version = trainer.process_batch() or version
            window.add_records_device(worker.collect_finished_device())
        else:
            window.add_records_device(worker.collect_finished_device())
        if trainer is not None and not (pipelined and not multi):
            version = trainer.process_batch() or version          # 0 while the window is still too small
            if log and rank == trainer_rank:
                log(f'version {version}: {len(window)} states in the window, {window.num_games} games')
        if multi and not data_parallel:
            v = torch.tensor([version], dtype=torch.int64, device=device)
            dist.broadcast(v, trainer_rank)
            version = int(v.item())
    return server, worker, trainer
